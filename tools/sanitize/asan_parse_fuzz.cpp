// Host-side hardening run (CPU only): mjx_parse + mjx_validate over mutated fixtures under AddressSanitizer / UBSan.
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -Iinclude -Ijpeg-rust_amd/csrc \
//       tools/sanitize/asan_parse_fuzz.cpp jpeg-rust_amd/csrc/mjx_parse.cpp jpeg-rust_amd/csrc/mjx_plan.cpp \
//       jpeg-rust_amd/csrc/mjx_lut.cpp -o /tmp/asan_parse_fuzz && /tmp/asan_parse_fuzz 400 tests/golden/pil/*.jpg tests/data/*
// (first argument: mutations per file; built and run by tests/test_sanitizers.py)
#include "mjx.h"
#include "mjx_plan.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

extern "C" int mjx_validate(const mjx_scan_desc *desc, const mjx_opts *opts)      // (the one in mjx_api.hip needs HIP)
{
    mjx::ImagePlan p;
    std::vector<mjx::ImagePlan> all;
    mjx_opts o{};
    if (opts) o = *opts;
    mjx::plan_input(*desc, o, all);
    return all.back().status;
}

int main(int argc, char **argv)
{
    std::mt19937_64 rng(12345);
    long runs = 0, ok = 0;
    const int per_file = argc > 1 ? std::atoi(argv[1]) : 400;
    for (int a = 2; a < argc; a++) {
        FILE *f = std::fopen(argv[a], "rb");
        if (!f) continue;
        std::vector<uint8_t> base;
        uint8_t buf[65536];
        size_t n;
        while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) base.insert(base.end(), buf, buf + n);
        std::fclose(f);
        if (base.size() < 4 || base.size() > (1u << 20)) continue;
        for (int k = 0; k < per_file; k++) {
            std::vector<uint8_t> b = base;
            const int muts = int(rng() % 8);
            for (int m = 0; m < muts; m++) {
                static const uint8_t pick[] = {0xff, 0xd0, 0xd7, 0xd9, 0xda, 0xc4, 0xdd, 0x00};
                b[rng() % b.size()] = (rng() & 1) ? pick[rng() % sizeof pick] : uint8_t(rng());
            }
            if (rng() % 3 == 0) b.resize(1 + rng() % b.size());
            for (int mode = 0; mode < 4; mode++) {
                mjx_opts o{};
                o.strict_ref = mode & 1;
                o.layout = (mode >> 1) & 1;
                o.device_destuff = (k & 1);
                mjx_scan_desc d;
                // exact-size heap copy: reads past the end are caught
                uint8_t *heap = static_cast<uint8_t *>(std::malloc(b.size()));
                std::memcpy(heap, b.data(), b.size());
                const int rc = mjx_parse(heap, b.size(), &o, &d);
                runs++;
                if (rc == MJX_OK) {
                    if (!d.scan_is_stuffed) ok += mjx_validate(&d, &o) == MJX_OK;
                    mjx_free_scan(&d);
                }
                std::free(heap);
            }
        }
    }
    std::printf("asan parse fuzz: %ld parses, %ld valid plans\n", runs, ok);
    return 0;
}
