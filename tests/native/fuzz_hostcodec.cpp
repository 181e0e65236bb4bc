// Sanitizer harness for the host-only codec behind the C ABI (csrc/hostcodec.hip has no device code, so it compiles with g++):
// built with -fsanitize=address,undefined by tests/test_tiff_cpu.py and run for a bounded number of cases.  Every buffer is a heap
// block of exactly the size passed as cap / n, so a one-byte overrun is an ASan report.  Uploaded GeoTIFFs reach the decoder
// (app/sr_routes.py /api/enhance), which is why garbage and mutated streams are part of the run.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../include/s2sr.h"

static uint64_t rs = 0x9E3779B97F4A7C15ull;
static uint32_t rnd() { rs ^= rs << 7; rs ^= rs >> 9; return (uint32_t)(rs >> 16); }

static bool same(const uint8_t* a, const uint8_t* b, size_t n) { return n == 0 || memcmp(a, b, n) == 0; }

static std::vector<uint8_t> make(int kind, size_t n) {
    std::vector<uint8_t> v(n);
    switch (kind) {
    case 0: for (auto& b : v) b = (uint8_t)rnd(); break;                                   // incompressible: the table fills fast
    case 1: for (size_t i = 0; i < n; ++i) v[i] = (uint8_t)(i / 97); break;                // long runs (KwKwK strings)
    case 2: for (auto& b : v) b = (uint8_t)(rnd() % 3); break;                             // tiny alphabet: long strings
    case 3: for (auto& b : v) b = 0; break;                                                 // one run: longest strings of all
    default: for (size_t i = 0; i < n; ++i) v[i] = (uint8_t)(128 + 60 * ((i >> 4) & 1) + rnd() % 5); break;   // image-like
    }
    return v;
}

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "FAIL %s:%d %s (case %d)\n", __FILE__, __LINE__, #c, ncase); return 1; } } while (0)

int main(int argc, char** argv) {
    const int cases = argc > 1 ? atoi(argv[1]) : 400;
    int ncase = 0;
    for (; ncase < cases; ++ncase) {
        const int kind = ncase % 5;
        const size_t n = ncase < 8 ? (size_t)ncase : (size_t)(rnd() % (ncase % 17 == 0 ? 300000 : 9000));
        std::vector<uint8_t> src = make(kind, n);
        const size_t cap = n * 3 / 2 + 16;
        uint8_t* enc = (uint8_t*)malloc(cap);
        size_t en = 0;
        CHECK(s2sr_tiff_lzw_encode(src.data(), n, enc, cap, &en) == S2SR_OK && en <= cap);
        // round trip into a block of exactly n bytes (n == 0: one byte, cap 0)
        uint8_t* dec = (uint8_t*)malloc(n ? n : 1);
        size_t dn = 0;
        CHECK(s2sr_tiff_lzw_decode(enc, en, dec, n, &dn) == S2SR_OK && dn == n && same(dec, src.data(), n));
        // more room than data: stops at EndOfInformation
        uint8_t* big = (uint8_t*)malloc(n + 100);
        CHECK(s2sr_tiff_lzw_decode(enc, en, big, n + 100, &dn) == S2SR_OK && dn == n);
        // less room: a prefix, never past cap
        const size_t part = n ? rnd() % n : 0;
        uint8_t* small = (uint8_t*)malloc(part ? part : 1);
        CHECK(s2sr_tiff_lzw_decode(enc, en, small, part, &dn) == S2SR_OK && dn == part && same(small, src.data(), part));
        // encoder without room: an error, not an overrun
        if (en > 2) {
            const size_t ecap = rnd() % en;
            uint8_t* e2 = (uint8_t*)malloc(ecap ? ecap : 1);
            size_t e2n = 0;
            CHECK(s2sr_tiff_lzw_encode(src.data(), n, e2, ecap, &e2n) == S2SR_E_CAPACITY);
            free(e2);
        }
        // truncated and bit-flipped streams, and plain garbage: OK or INVALID, out_n <= cap, nothing read past the stream
        for (int m = 0; m < 4; ++m) {
            size_t mn = m == 0 ? (en ? rnd() % en : 0) : en;
            uint8_t* mut = (uint8_t*)malloc(mn ? mn : 1);
            if (mn) memcpy(mut, enc, mn);
            if (m == 3) for (size_t i = 0; i < mn; ++i) mut[i] = (uint8_t)rnd();
            else if (m > 0 && mn) for (int f = 0; f < m * 3; ++f) mut[rnd() % mn] ^= (uint8_t)(1u << (rnd() % 8));
            const size_t gcap = rnd() % (n + 64);
            uint8_t* g = (uint8_t*)malloc(gcap ? gcap : 1);
            size_t gn = (size_t)-1;
            const int rc = s2sr_tiff_lzw_decode(mut, mn, g, gcap, &gn);
            CHECK(rc == S2SR_OK || rc == S2SR_E_INVALID);
            CHECK(rc != S2SR_OK || gn <= gcap);
            free(g);
            free(mut);
        }
        free(small); free(big); free(dec); free(enc);
    }
    // null arguments
    size_t k = 0;
    uint8_t b[4] = {0};
    CHECK(s2sr_tiff_lzw_decode(nullptr, 3, b, 4, &k) == S2SR_E_INVALID && s2sr_tiff_lzw_encode(b, 4, nullptr, 0, &k) == S2SR_E_INVALID);
    printf("ok %d cases\n", ncase);
    return 0;
}
