// The decode algorithm itself (tests/emul: the per-lane routine of the kernels, run on the CPU) under AddressSanitizer /
// UBSan over files with mutated Huffman tables and scan bytes: table lookups, links and stream writes must stay in bounds.
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -Iinclude -Ijpeg-rust_amd/csrc \
//       tools/sanitize/asan_emul_fuzz.cpp tests/emul/huff_emul.cpp jpeg-rust_amd/csrc/mjx_parse.cpp jpeg-rust_amd/csrc/mjx_plan.cpp \
//       jpeg-rust_amd/csrc/mjx_lut.cpp -o /tmp/asan_emul_fuzz && /tmp/asan_emul_fuzz 60 tests/golden/pil/opt_*.jpg tests/data/*.jp*
// (first argument: mutations per file; built and run by tests/test_sanitizers.py)
#include "mjx.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
extern "C" int emul_decode_coefs(const uint8_t *jpeg, size_t len, int layout, int mode, int16_t *out, size_t cap_blocks, size_t *nblocks, int *stats);
int main(int argc, char **argv)
{
    std::mt19937_64 rng(777);
    long runs = 0, ok = 0;
    std::vector<int16_t> out(size_t(200000) * 64);
    const int per_file = argc > 1 ? std::atoi(argv[1]) : 60;
    for (int a = 2; a < argc; a++) {
        FILE *f = std::fopen(argv[a], "rb");
        if (!f) continue;
        std::vector<uint8_t> base; uint8_t buf[65536]; size_t n;
        while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) base.insert(base.end(), buf, buf + n);
        std::fclose(f);
        if (base.size() < 4 || base.size() > 120000) continue;
        // locate DHT payloads
        std::vector<std::pair<size_t,size_t>> dht;
        for (size_t i = 2; i + 4 < base.size();) {
            if (base[i] != 0xff) break;
            const uint8_t m = base[i+1]; const size_t ln = (size_t(base[i+2]) << 8) | base[i+3];
            if (m == 0xc4) dht.push_back({i + 4, ln - 2});
            if (m == 0xda) break;
            i += 2 + ln;
        }
        for (int k = 0; k < per_file; k++) {
            std::vector<uint8_t> b = base;
            const int muts = 1 + int(rng() % 4);
            for (int m = 0; m < muts; m++) {
                if (!dht.empty() && (rng() & 1)) { auto d = dht[rng() % dht.size()]; b[d.first + rng() % d.second] = uint8_t(rng()); }
                else b[rng() % b.size()] = uint8_t(rng());
            }
            size_t nb = 0; int st[8];
            uint8_t *heap = static_cast<uint8_t *>(std::malloc(b.size()));
            std::memcpy(heap, b.data(), b.size());
            const int rc = emul_decode_coefs(heap, b.size(), 0, int(k & 1), out.data(), 200000, &nb, st);
            std::free(heap);
            runs++; ok += rc == 0;
        }
    }
    std::printf("asan emul fuzz: %ld decodes, %ld clean\n", runs, ok);
}
