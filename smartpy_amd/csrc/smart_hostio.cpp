// smart_hostio.cpp -- host side of the sampling database (include/smart_amd.h, "sampling database" section).
//
// The reference appends one text line per sample from Python (montecarlo.py:211-231): every value cast to float32
// and printed '%.6e', comma separated, '\n' terminated.  At 1e5 samples the GPU run takes 13 ms and that loop (or
// numpy.savetxt standing in for it) 0.4-1 s, so the rows are formatted here: worker threads over blocks of rows,
// written in order.  The characters are those of CPython's '%.6e' % float (checked against Python in
// tests/test_host_logic.py): a double-precision fast path whose error bound decides all but the near-ties, and
// glibc's correctly rounded snprintf for those.
#include "../../include/smart_amd.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

namespace smart {
int fail(int code, const char *fmt, ...); // smart_capi.hip: sets the text smart_last_error() returns
}

namespace {

constexpr int kCell = 14; // "-d.dddddde+XX" is 13 characters, plus the separator

// 10^n for n = -32 .. 51 (the scalings a float32 needs to bring its 7 leading digits in front of the point)
constexpr int kPowLo = -32, kPowHi = 51;
const double kPow10[kPowHi - kPowLo + 1] = {
    1e-32, 1e-31, 1e-30, 1e-29, 1e-28, 1e-27, 1e-26, 1e-25, 1e-24, 1e-23, 1e-22, 1e-21, 1e-20, 1e-19, 1e-18, 1e-17,
    1e-16, 1e-15, 1e-14, 1e-13, 1e-12, 1e-11, 1e-10, 1e-9,  1e-8,  1e-7,  1e-6,  1e-5,  1e-4,  1e-3,  1e-2,  1e-1,
    1e0,   1e1,   1e2,   1e3,   1e4,   1e5,   1e6,   1e7,   1e8,   1e9,   1e10,  1e11,  1e12,  1e13,  1e14,  1e15,
    1e16,  1e17,  1e18,  1e19,  1e20,  1e21,  1e22,  1e23,  1e24,  1e25,  1e26,  1e27,  1e28,  1e29,  1e30,  1e31,
    1e32,  1e33,  1e34,  1e35,  1e36,  1e37,  1e38,  1e39,  1e40,  1e41,  1e42,  1e43,  1e44,  1e45,  1e46,  1e47,
    1e48,  1e49,  1e50,  1e51};

// One value the way Python prints '%.6e' % numpy.float32(v): the float32 widened to double, 7 significant digits,
// correctly rounded (ties to even on the exact value).  Fast path: x = |v| * 10^(6 - k) in double carries a relative
// error <= 2.3e-16, i.e. < 3e-9 on the 7-digit integer; unless x lies within 1e-5 of a rounding tie or of a decade
// boundary the digits are decided, otherwise glibc's exact snprintf decides (about one value in 50,000).
inline char *put_value(char *p, float v)
{
    if (std::isnan(v)) { // glibc would print "-nan" for a negative NaN
        memcpy(p, "nan", 3);
        return p + 3;
    }
    const double a = std::fabs((double)v);
    if (a == 0.0 || std::isinf(v))
        return p + snprintf(p, kCell + 8, "%.6e", (double)v);
    int e2;
    (void)std::frexp(a, &e2);
    int k = (int)std::floor((e2 - 1) * 0.30102999566398120); // floor(log10(a)) or one less
    double x = a * kPow10[6 - k - kPowLo];
    if (x >= 1e7) {
        ++k;
        x = a * kPow10[6 - k - kPowLo];
    }
    const double fl = std::floor(x), frac = x - fl;
    if (std::fabs(frac - 0.5) < 1e-5 || x < 1e6 + 1e-5 || x > 1e7 - 1e-5)
        return p + snprintf(p, kCell + 8, "%.6e", (double)v);
    long d = (long)fl + (frac > 0.5 ? 1 : 0);
    if (d == 10000000) {
        d = 1000000;
        ++k;
    }
    if (std::signbit(v))
        *p++ = '-';
    char digits[7];
    for (int i = 6; i >= 0; --i) {
        digits[i] = (char)('0' + d % 10);
        d /= 10;
    }
    *p++ = digits[0];
    *p++ = '.';
    memcpy(p, digits + 1, 6);
    p += 6;
    *p++ = 'e';
    *p++ = k < 0 ? '-' : '+';
    const int ak = k < 0 ? -k : k;
    *p++ = (char)('0' + ak / 10);
    *p++ = (char)('0' + ak % 10);
    return p;
}

size_t format_rows(const float *table, int64_t r0, int64_t r1, int64_t n_cols, char *out)
{
    char *p = out;
    for (int64_t r = r0; r < r1; ++r) {
        const float *row = table + r * n_cols;
        for (int64_t c = 0; c < n_cols; ++c) {
            p = put_value(p, row[c]);
            *p++ = c + 1 < n_cols ? ',' : '\n';
        }
    }
    return (size_t)(p - out);
}

} // namespace

extern "C" int smart_db_append_rows(const char *path, const float *table, int64_t n_rows, int64_t n_cols,
                                    int32_t n_threads)
{
    if (!path || (!table && n_rows > 0) || n_rows < 0 || n_cols < 1)
        return smart::fail(SMART_E_SIZE, "smart_db_append_rows: path, table and sizes are required");
    FILE *f = fopen(path, "ab");
    if (!f)
        return smart::fail(SMART_E_IO, "smart_db_append_rows: cannot open '%s' for appending", path);
    int rc = SMART_OK;
    if (n_rows > 0) {
        int hw = (int)std::thread::hardware_concurrency();
        int nt = n_threads > 0 ? n_threads : std::min(hw > 0 ? hw : 1, 16);
        const int64_t rows_per_block = std::max<int64_t>(1, (int64_t)(1 << 16) / n_cols); // ~1 MB of text per block
        const int64_t n_blocks = (n_rows + rows_per_block - 1) / rows_per_block;
        nt = (int)std::min<int64_t>(nt, n_blocks);
        const size_t block_cap = (size_t)rows_per_block * (size_t)n_cols * kCell + 16;
        std::vector<std::vector<char>> buf((size_t)nt);
        std::vector<size_t> len((size_t)nt);
        for (auto &b : buf)
            b.resize(block_cap);
        // rounds of nt blocks: formatted in parallel, written in order
        for (int64_t b0 = 0; b0 < n_blocks && rc == SMART_OK; b0 += nt) {
            const int in_round = (int)std::min<int64_t>(nt, n_blocks - b0);
            std::vector<std::thread> workers;
            for (int t = 1; t < in_round; ++t)
                workers.emplace_back([&, t] {
                    const int64_t r0 = (b0 + t) * rows_per_block;
                    len[t] = format_rows(table, r0, std::min(n_rows, r0 + rows_per_block), n_cols, buf[t].data());
                });
            len[0] = format_rows(table, b0 * rows_per_block, std::min(n_rows, (b0 + 1) * rows_per_block), n_cols,
                                 buf[0].data());
            for (auto &w : workers)
                w.join();
            for (int t = 0; t < in_round; ++t)
                if (fwrite(buf[t].data(), 1, len[t], f) != len[t]) {
                    rc = SMART_E_IO;
                    break;
                }
        }
    }
    if (fclose(f) != 0)
        rc = SMART_E_IO;
    if (rc != SMART_OK)
        return smart::fail(rc, "smart_db_append_rows: writing '%s' failed", path);
    return rc;
}

// ---- reading the rows back (GLUE / Best / Total read the LHS database, montecarlo.py:233-262) ------------------
// text: the file's bytes after the header line.  Every line has n_cols comma-separated values; the `n_use` columns
// listed in `cols` are parsed with strtod (correctly rounded to double) and narrowed to float32 -- the route
// numpy.array(list_of_str, dtype=float32) takes in the reference.  Returns the number of rows, or a negative code.
extern "C" int64_t smart_db_parse_rows(const char *text, int64_t len, int64_t n_cols, const int32_t *cols,
                                       int32_t n_use, float *out, int64_t max_rows, int32_t n_threads)
{
    if (!text || len < 0 || n_cols < 1 || !cols || n_use < 1 || (!out && max_rows > 0))
        return smart::fail(SMART_E_SIZE, "smart_db_parse_rows: text, columns and output are required");
    for (int32_t j = 0; j < n_use; ++j)
        if (cols[j] < 0 || cols[j] >= n_cols)
            return smart::fail(SMART_E_SIZE, "smart_db_parse_rows: column %d outside 0..%lld", (int)cols[j],
                               (long long)n_cols - 1);
    // line starts (empty lines are skipped, a trailing '\r' is tolerated)
    std::vector<std::pair<const char *, const char *>> lines;
    for (const char *p = text, *end = text + len; p < end;) {
        const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
        const char *stop = nl ? nl : end;
        const char *q = stop;
        if (q > p && q[-1] == '\r')
            --q;
        if (q > p)
            lines.emplace_back(p, q);
        p = stop + 1;
    }
    const int64_t n_rows = (int64_t)lines.size();
    if (n_rows > max_rows)
        return smart::fail(SMART_E_SIZE, "smart_db_parse_rows: %lld rows, room for %lld", (long long)n_rows,
                           (long long)max_rows);
    // where does each wanted column go in the output row?
    std::vector<int32_t> slot((size_t)n_cols, -1);
    for (int32_t j = 0; j < n_use; ++j)
        slot[(size_t)cols[j]] = j; // a column asked for twice is filled once, below
    int hw = (int)std::thread::hardware_concurrency();
    int nt = n_threads > 0 ? n_threads : std::min(hw > 0 ? hw : 1, 16);
    nt = (int)std::max<int64_t>(1, std::min<int64_t>(nt, n_rows / 256));
    std::vector<int64_t> bad((size_t)nt, -1);
    auto work = [&](int t) {
        const int64_t r0 = n_rows * t / nt, r1 = n_rows * (t + 1) / nt;
        char cell[64];
        for (int64_t r = r0; r < r1; ++r) {
            const char *p = lines[(size_t)r].first, *end = lines[(size_t)r].second;
            float *o = out + r * n_use;
            int64_t c = 0;
            while (c < n_cols) {
                const char *comma = (const char *)memchr(p, ',', (size_t)(end - p));
                const char *stop = comma ? comma : end;
                if (slot[(size_t)c] >= 0) {
                    const size_t w = (size_t)(stop - p);
                    if (w == 0 || w >= sizeof(cell)) {
                        bad[(size_t)t] = r;
                        return;
                    }
                    memcpy(cell, p, w);
                    cell[w] = 0;
                    char *done = nullptr;
                    const double v = strtod(cell, &done);
                    if (done != cell + w) {
                        bad[(size_t)t] = r;
                        return;
                    }
                    o[slot[(size_t)c]] = (float)v;
                }
                ++c;
                if (!comma)
                    break;
                p = comma + 1;
            }
            if (c != n_cols) { // too few values on the line
                bad[(size_t)t] = r;
                return;
            }
        }
    };
    std::vector<std::thread> workers;
    for (int t = 1; t < nt; ++t)
        workers.emplace_back(work, t);
    work(0);
    for (auto &w : workers)
        w.join();
    for (int t = 0; t < nt; ++t)
        if (bad[(size_t)t] >= 0)
            return smart::fail(SMART_E_IO, "smart_db_parse_rows: line %lld of the table is malformed",
                               (long long)bad[(size_t)t] + 1);
    for (int32_t j = 0; j < n_use; ++j) // duplicates: copy from the slot that was filled
        if (slot[(size_t)cols[j]] != j)
            for (int64_t r = 0; r < n_rows; ++r)
                out[r * n_use + j] = out[r * n_use + slot[(size_t)cols[j]]];
    return n_rows;
}
