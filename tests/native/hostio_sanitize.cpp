// hostio_sanitize.cpp -- the host-side native code of the library (smart_hostio.cpp: the sampling-database row writer
// and parser) under AddressSanitizer + UndefinedBehaviourSanitizer, on the CPU.  (GPU sanitizers are not available on
// this pool; the kernels are covered by the parity tests.)  Built and run by tests/test_host_logic.py.
//
// Random float32 tables, including specials, go through smart_db_append_rows (several thread counts) into a file and
// back through smart_db_parse_rows; the text must equal snprintf("%.6e") (Python's spelling of NaN) and the parsed values must equal strtod ->
// float.  Malformed input, missing room and bad column indices must come back as error codes.
#include "../../include/smart_amd.h"

#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

namespace smart {
int fail(int code, const char *fmt, ...) // the library's own is in smart_capi.hip (needs the HIP runtime)
{
    va_list ap;
    va_start(ap, fmt);
    vfprintf(stderr, fmt, ap);
    fputc('\n', stderr);
    va_end(ap);
    return code;
}
} // namespace smart

#define REQUIRE(cond)                                                                                                  \
    do {                                                                                                               \
        if (!(cond)) {                                                                                                 \
            fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, #cond);                                                 \
            return 1;                                                                                                  \
        }                                                                                                              \
    } while (0)

// what Python prints for '%.6e' % numpy.float32(v): C's "%.6e" of the double, except that a NaN has no sign
static void python_cell(float v, char *cell, size_t n)
{
    if (std::isnan(v))
        snprintf(cell, n, "nan");
    else
        snprintf(cell, n, "%.6e", (double)v);
}

static std::string slurp(const char *path)
{
    std::string s;
    FILE *f = fopen(path, "rb");
    char buf[65536];
    size_t n;
    while (f && (n = fread(buf, 1, sizeof(buf), f)) > 0)
        s.append(buf, n);
    if (f)
        fclose(f);
    return s;
}

int main(int argc, char **argv)
{
    const char *path = argc > 1 ? argv[1] : "/tmp/hostio_sanitize.csv";
    std::mt19937_64 rng(12345);
    for (int round = 0; round < 6; ++round) {
        const int64_t rows = round == 0 ? 1 : 1 + (int64_t)(rng() % 3000), cols = 1 + (int64_t)(rng() % 40);
        std::vector<float> table((size_t)(rows * cols));
        for (auto &v : table) {
            const uint32_t bits = (uint32_t)rng();
            memcpy(&v, &bits, 4);
            if (rng() % 50 == 0)
                v = (rng() & 1) ? 0.0f : -0.0f;
        }
        if (table.size() > 4) {
            table[1] = INFINITY, table[2] = -INFINITY, table[3] = NAN, table[4] = 1e-45f;
        }
        remove(path);
        REQUIRE(smart_db_append_rows(path, table.data(), rows, cols, round % 5) == SMART_OK);
        const std::string text = slurp(path);
        std::string want;
        char cell[64];
        for (int64_t r = 0; r < rows; ++r)
            for (int64_t c = 0; c < cols; ++c) {
                python_cell(table[(size_t)(r * cols + c)], cell, sizeof(cell));
                want += cell;
                want += c + 1 == cols ? '\n' : ',';
            }
        REQUIRE(text == want);
        std::vector<int32_t> pick;
        for (int64_t c = cols - 1; c >= 0; c -= 2)
            pick.push_back((int32_t)c);
        std::vector<float> back((size_t)(rows * (int64_t)pick.size()), -7.0f);
        const int64_t got = smart_db_parse_rows(text.data(), (int64_t)text.size(), cols, pick.data(), (int32_t)pick.size(),
                                                back.data(), rows, 1 + round % 4);
        REQUIRE(got == rows);
        for (int64_t r = 0; r < rows; ++r)
            for (size_t j = 0; j < pick.size(); ++j) {
                python_cell(table[(size_t)(r * cols + pick[j])], cell, sizeof(cell));
                const float expect = (float)strtod(cell, nullptr);
                const float have = back[(size_t)r * pick.size() + j];
                REQUIRE((std::isnan(expect) && std::isnan(have)) || memcmp(&expect, &have, 4) == 0);
            }
        // error paths: no room, a column that is not there, a broken separator, an empty table
        REQUIRE(smart_db_parse_rows(text.data(), (int64_t)text.size(), cols, pick.data(), (int32_t)pick.size(), back.data(),
                                    rows - 1, 2) < 0 || rows == 0);
        const int32_t beyond = (int32_t)cols;
        REQUIRE(smart_db_parse_rows(text.data(), (int64_t)text.size(), cols, &beyond, 1, back.data(), rows, 2) < 0);
        if (cols > 1) {
            std::string broken = text;
            broken[broken.find(',')] = ';';
            REQUIRE(smart_db_parse_rows(broken.data(), (int64_t)broken.size(), cols, pick.data(), (int32_t)pick.size(),
                                        back.data(), rows, 3) < 0);
        }
        REQUIRE(smart_db_parse_rows("", 0, cols, pick.data(), (int32_t)pick.size(), back.data(), rows, 2) == 0);
    }
    REQUIRE(smart_db_append_rows("/nonexistent-directory/x.csv", nullptr, 1, 1, 1) < 0);
    remove(path);
    puts("hostio sanitize ok");
    return 0;
}
