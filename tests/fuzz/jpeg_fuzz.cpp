// Test infrastructure (CPU only): mutation fuzzing of the host JPEG entropy decoder under AddressSanitizer + UBSan.
// Built by tests/test_jpeg_fuzz.py as   g++ -fsanitize=address,undefined jpeg_fuzz.cpp ../../vip-cup-2022_amd/csrc/jpeg_host.cpp
// The decoder takes untrusted files (main.py reads whatever the CSV names): whatever the bytes are, it must return a
// status - never read or write outside its buffers.  usage: jpeg_fuzz <iterations per file> <file.jpg>...
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "vipcup_hip.h"

void vip_set_error(const char*, ...) {}

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd() {
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return (uint32_t)(rng_state >> 16);
}

static int run_one(const std::vector<uint8_t>& buf, long* decoded) {
    vip_jpeg_desc d;
    size_t elems = 0;
    // exact-size heap copy: ASan sees a read one byte past the end of the stream
    std::vector<uint8_t> copy(buf);
    const uint8_t* p = copy.data();
    size_t len = copy.size();
    if (vip_jpeg_probe_h(p, len, &d, &elems) != VIP_OK) return 0;
    if (elems == 0 || elems > (size_t)64 << 20) return 0;      // a mutated header may ask for gigabytes: not a decoder bug
    std::vector<int16_t> coef(elems);
    size_t used = 0;
    const int st = vip_jpeg_entropy_decode_h(&p, &len, 1, &d, coef.data(), elems, &used, 1);
    if (st == VIP_OK) ++*decoded;
    return 0;
}

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    const int iters = atoi(argv[1]);
    long total = 0, decoded = 0;
    for (int f = 2; f < argc; ++f) {
        FILE* fp = fopen(argv[f], "rb");
        if (!fp) return 3;
        std::vector<uint8_t> orig;
        uint8_t tmp[65536];
        size_t n;
        while ((n = fread(tmp, 1, sizeof tmp, fp)) > 0) orig.insert(orig.end(), tmp, tmp + n);
        fclose(fp);
        run_one(orig, &decoded);
        ++total;
        for (int it = 0; it < iters; ++it) {
            std::vector<uint8_t> m(orig);
            switch (rnd() % 5) {
                case 0: m.resize(rnd() % (m.size() + 1)); break;                           // truncate anywhere
                case 1: for (int k = 1 + rnd() % 8; k > 0; --k) m[rnd() % m.size()] ^= (uint8_t)(1u << (rnd() % 8)); break;
                case 2: for (int k = 1 + rnd() % 4; k > 0; --k) m[rnd() % m.size()] = 0xFF; break;   // spurious markers
                case 3: {                                                                  // header bytes only
                    const size_t hdr = m.size() < 700 ? m.size() : 700;
                    for (int k = 1 + rnd() % 6; k > 0; --k) m[rnd() % hdr] = (uint8_t)rnd();
                    break;
                }
                default: {                                                                 // cut a span out of the middle
                    const size_t a = rnd() % m.size(), b = a + rnd() % (m.size() - a + 1);
                    m.erase(m.begin() + a, m.begin() + b);
                    break;
                }
            }
            if (m.empty()) continue;
            run_one(m, &decoded);
            ++total;
        }
    }
    printf("fuzzed %ld streams, %ld decoded to the end\n", total, decoded);
    return 0;
}
