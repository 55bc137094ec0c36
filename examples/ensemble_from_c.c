/*
 * ensemble_from_c.c -- the C ABI of include/smart_amd.h driven from plain C: no Python, no torch.
 *
 * What a compiled caller (the native side of the reference's `smartcpp` hook, or any C / C++ / Fortran host code)
 * does to run a Monte-Carlo ensemble: put parameters, forcing and observations into device memory, ask the library
 * how much scratch it wants, plan, launch, read the status word, copy the results back.  The caller owns every
 * buffer; the library allocates nothing.
 *
 *   gcc -O2 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/ensemble_from_c.c \
 *       -Lsmartpy_amd/csrc -lsmart_amd -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,$PWD/smartpy_amd/csrc -o ensemble_from_c
 *   ./ensemble_from_c params.bin forcing.bin obs.bin N T W gap area out.bin
 *
 * Inputs are raw little-endian doubles: params [N][10], forcing [T][2], obs [T / gap] (NaN = missing).  Output:
 * objfn [N][8], then gw [N], then discharge [T / gap][N] (sample-minor, as the kernel writes it).
 * tests/test_gpu_api.py::test_the_c_abi_from_a_c_program checks the numbers against the Python binding, bit for bit.
 */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "smart_amd.h"

#define CHECK_HIP(call)                                                                                                \
    do {                                                                                                               \
        hipError_t e_ = (call);                                                                                        \
        if (e_ != hipSuccess) {                                                                                        \
            fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_));                                                 \
            return 2;                                                                                                  \
        }                                                                                                              \
    } while (0)

#define CHECK_SMART(call)                                                                                              \
    do {                                                                                                               \
        int rc_ = (call);                                                                                              \
        if (rc_ != SMART_OK) {                                                                                         \
            fprintf(stderr, "%s: error %d: %s\n", #call, rc_, smart_last_error());                                     \
            return 3;                                                                                                  \
        }                                                                                                              \
    } while (0)

static double *read_doubles(const char *path, size_t n)
{
    double *buf = (double *)malloc(n * sizeof(double));
    FILE *f = fopen(path, "rb");
    if (!buf || !f || fread(buf, sizeof(double), n, f) != n) {
        fprintf(stderr, "cannot read %zu doubles from %s\n", n, path);
        exit(1);
    }
    fclose(f);
    return buf;
}

static double *to_device(const double *host, size_t n)
{
    double *dev = NULL;
    if (hipMalloc((void **)&dev, n * sizeof(double)) != hipSuccess ||
        hipMemcpy(dev, host, n * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) {
        fprintf(stderr, "hipMalloc / hipMemcpy of %zu doubles failed\n", n);
        exit(2);
    }
    return dev;
}

int main(int argc, char **argv)
{
    if (argc != 10) {
        fprintf(stderr, "usage: %s params.bin forcing.bin obs.bin N T W gap area out.bin\n", argv[0]);
        return 1;
    }
    const int64_t N = atoll(argv[4]), T = atoll(argv[5]), W = atoll(argv[6]), gap = atoll(argv[7]);
    const double area = atof(argv[8]);
    const int64_t R = smart_n_reports(T, gap, SMART_REPORT_SUMMARY);
    if (smart_abi_version() != SMART_AMD_ABI_VERSION || smart_device_count() < 1) {
        fprintf(stderr, "library ABI %d (header %d), %d HIP device(s)\n", smart_abi_version(), SMART_AMD_ABI_VERSION,
                smart_device_count());
        return 1;
    }
    double *params = read_doubles(argv[1], (size_t)N * 10), *forcing = read_doubles(argv[2], (size_t)T * 2);
    double *obs = read_doubles(argv[3], (size_t)R);
    const double extra[7] = {1200.0, 0.45, 0.10, 0.15, 0.15, 0.30, 0.30}; /* aar, r-o_ratio, r-o_split[5] */
    const double gw_obs = 0.12667;

    SmartEnsemble e;
    memset(&e, 0, sizeof(e)); /* zero = the library decides (time slices, plan) */
    e.n_catchments = 1;
    e.n_samples = N;
    e.n_steps = T;
    e.n_warm = W;
    e.report_gap = gap;
    e.report_type = SMART_REPORT_SUMMARY;
    e.math_mode = SMART_MATH_FAST;
    e.delta_sec = 3600.0;
    e.area_m2 = to_device(&area, 1);
    e.forcing = to_device(forcing, (size_t)T * 2);
    e.params = to_device(params, (size_t)N * 10);
    e.extra = to_device(extra, 7);
    e.obs = to_device(obs, (size_t)R);
    e.gw_obs = to_device(&gw_obs, 1);
    const size_t n_out = (size_t)N * 8 + (size_t)N + (size_t)R * (size_t)N;
    double *out_dev = NULL;
    CHECK_HIP(hipMalloc((void **)&out_dev, n_out * sizeof(double)));
    e.objfn = out_dev;
    e.gw = out_dev + (size_t)N * 8;
    e.discharge = e.gw + N;
    e.discharge_ld = N;
    /* the objective functions need their scratch before the size query counts it */
    e.workspace_bytes = smart_workspace_bytes(&e);
    CHECK_HIP(hipMalloc(&e.workspace, (size_t)e.workspace_bytes));
    hipStream_t stream;
    CHECK_HIP(hipStreamCreate(&stream));
    e.stream = stream;

    char what[256];
    int32_t plan = 0, status = 0;
    CHECK_SMART(smart_check_ensemble(&e));
    CHECK_SMART(smart_plan_ensemble(&e, &plan)); /* which kernels do these rows and this forcing need? */
    e.plan = plan;
    CHECK_SMART(smart_describe_launch(&e, what, sizeof(what)));
    CHECK_SMART(smart_run_ensemble_hip(&e));     /* asynchronous on e.stream */
    CHECK_SMART(smart_launch_status(&e, &status)); /* waits for the stream */
    if (status & SMART_STATUS_SLICE_TIMEOUT) {   /* never seen; the documented answer is one unsliced repeat */
        e.time_slices = 1;
        CHECK_SMART(smart_run_ensemble_hip(&e));
        CHECK_SMART(smart_launch_status(&e, &status));
    }
    if (status != 0) {
        fprintf(stderr, "launch status %d\n", status);
        return 4;
    }
    double *out = (double *)malloc(n_out * sizeof(double));
    CHECK_HIP(hipMemcpy(out, out_dev, n_out * sizeof(double), hipMemcpyDeviceToHost));
    FILE *f = fopen(argv[9], "wb");
    if (!f || fwrite(out, sizeof(double), n_out, f) != n_out) {
        fprintf(stderr, "cannot write %s\n", argv[9]);
        return 1;
    }
    fclose(f);
    printf("%s | plan 0x%x | workspace %lld B | NSE of sample 0: %.6f | gw of sample 0: %.6f\n", what, plan,
           (long long)e.workspace_bytes, out[0], out[(size_t)N * 8]);
    return 0;
}
