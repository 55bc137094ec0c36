// placement.hip -- where does the dispatcher put G one-wave workgroups of a kernel that uses V VGPRs?
// Each wave spins for a fixed number of fp64 FMAs and records (XCC id, HW_ID, start, end).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
#include <algorithm>

#define SPIN_KERNEL(NV)                                                                                               \
    __global__ __launch_bounds__(64) __attribute__((amdgpu_num_vgpr(NV))) void spin##NV(unsigned long long *rec,         \
                                                                                         int iters, double *sink)        \
    {                                                                                                                 \
        spin_body(rec, iters, sink);                                                                                  \
    }

__device__ __forceinline__ void spin_body(unsigned long long *rec, int iters, double *sink)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    double x = threadIdx.x, a = 1.0000001, b = 1e-9;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j)
            asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(x) : "v"(a), "v"(b));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));    // HW_REG_HW_ID, 32 bits
        unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));  // HW_REG_XCC_ID
        rec[blockIdx.x * 4 + 0] = xcc;
        rec[blockIdx.x * 4 + 1] = hw;
        rec[blockIdx.x * 4 + 2] = t0;
        rec[blockIdx.x * 4 + 3] = t1;
    }
    if (x == 12345.678) sink[0] = x;
}

SPIN_KERNEL(64)
SPIN_KERNEL(128)
SPIN_KERNEL(144)
SPIN_KERNEL(256)

template <int NV, class K>
void run(K kernel, int grid, int iters)
{
    unsigned long long *rec;
    double *sink;
    hipMalloc(&rec, (size_t)grid * 4 * sizeof(unsigned long long));
    hipMalloc(&sink, 8);
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(64), 0, 0, rec, 10, sink);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(64), 0, 0, rec, iters, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h((size_t)grid * 4);
    hipMemcpy(h.data(), rec, h.size() * 8, hipMemcpyDeviceToHost);
    // group by (xcc, se, cu, simd): gfx9 HW_ID: wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh_id[12] se_id[15:13]
    std::map<unsigned long long, int> per_simd, per_cu;
    unsigned long long tmin = ~0ull, tmax = 0;
    for (int b = 0; b < grid; ++b) {
        unsigned xcc = (unsigned)h[b * 4] & 0xf, hw = (unsigned)h[b * 4 + 1];
        unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        unsigned long long cuk = ((unsigned long long)xcc << 16) | (se << 8) | (sh << 4) | cu;
        per_cu[cuk]++;
        per_simd[(cuk << 4) | simd]++;
        tmin = std::min(tmin, h[b * 4 + 2]);
        tmax = std::max(tmax, h[b * 4 + 3]);
    }
    std::map<int, int> hist_simd, hist_cu;
    for (auto &kv : per_simd) hist_simd[kv.second]++;
    for (auto &kv : per_cu) hist_cu[kv.second]++;
    // one wave alone: duration of the shortest wave
    unsigned long long dmin = ~0ull, dmax = 0;
    for (int b = 0; b < grid; ++b) {
        unsigned long long d = h[b * 4 + 3] - h[b * 4 + 2];
        dmin = std::min(dmin, d);
        dmax = std::max(dmax, d);
    }
    printf("VGPR=%3d grid=%5d  %.3f ms | SIMDs used %zu, CUs used %zu | waves/SIMD histogram:", NV, grid, ms, per_simd.size(), per_cu.size());
    for (auto &kv : hist_simd) printf(" %dw:%d", kv.first, kv.second);
    printf(" | waves/CU:");
    for (auto &kv : hist_cu) printf(" %d:%d", kv.first, kv.second);
    printf(" | wave duration min %.3f max %.3f ms (100 MHz ticks)\n", dmin / 1e5, dmax / 1e5);
    hipFree(rec);
    hipFree(sink);
}

int main(int argc, char **argv)
{
    const int iters = 40000;
    for (int grid : {1024, 1563, 1954, 2048, 3126}) {
        run<64>(spin64, grid, iters);
        run<128>(spin128, grid, iters);
        run<144>(spin144, grid, iters);
        run<256>(spin256, grid, iters);
    }
    return 0;
}
