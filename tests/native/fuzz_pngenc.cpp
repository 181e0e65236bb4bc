// Sanitizer harness for the host-only PNG encoder (csrc/pngenc.hip has no device code, so it compiles with g++): built with
// -fsanitize=address,undefined by tests/test_tiff_cpu.py.  Output buffers are heap blocks of exactly the capacity passed; every
// file is parsed chunk by chunk (lengths, CRCs), its IDAT stream inflated with zlib and un-filtered back to the input pixels.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include <vector>

#include "../../include/s2sr.h"

static uint64_t rs = 0x2545F4914F6CDD1Dull;
static uint32_t rnd() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (uint32_t)(rs >> 13); }
static uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | (p[1] << 16) | (p[2] << 8) | p[3]; }

static void fill(std::vector<uint8_t>& px, int w, int h, int c, int kind) {
    px.resize((size_t)w * h * c);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x)
            for (int k = 0; k < c; ++k) {
                uint8_t v;
                switch (kind) {
                case 0: v = (uint8_t)rnd(); break;                                        // noise: stored blocks
                case 1: v = 0; break;                                                     // one long run
                case 2: v = (uint8_t)((x / 6 + y / 6) * 3 + k * 40); break;               // upsampled: runs of 6 pixels
                case 3: v = (uint8_t)(x + y + (rnd() % 3)); break;                        // gradient + a little noise
                case 4: { uint32_t r = rnd(); int g = 0; while ((r & 1) && g < 30) { r >>= 1; ++g; } v = (uint8_t)(g * 9); } break;   // skewed: deep trees
                default: v = (uint8_t)((rnd() % 100 < 70) ? 7 : rnd()); break;            // runs broken by noise
                }
                px[((size_t)y * w + x) * c + k] = (k == 3 && kind != 0) ? 255 : v;
            }
}

// parse a PNG: returns the concatenated IDAT payload, checks chunk framing and CRCs
static bool parse(const uint8_t* f, size_t n, int w, int h, int c, std::vector<uint8_t>& idat) {
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
    if (n < 8 || memcmp(f, sig, 8)) return false;
    size_t pos = 8;
    bool ihdr = false, iend = false;
    while (pos + 12 <= n) {
        const uint32_t len = be32(f + pos);
        if (pos + 12 + len > n) return false;
        const uint32_t crc = (uint32_t)crc32(crc32(0, Z_NULL, 0), f + pos + 4, len + 4);
        if (crc != be32(f + pos + 8 + len)) return false;
        if (!memcmp(f + pos + 4, "IHDR", 4)) {
            if (len != 13 || (int)be32(f + pos + 8) != w || (int)be32(f + pos + 12) != h || f[pos + 16] != 8 || f[pos + 17] != (c == 3 ? 2 : 6)) return false;
            ihdr = true;
        } else if (!memcmp(f + pos + 4, "IDAT", 4)) idat.insert(idat.end(), f + pos + 8, f + pos + 8 + len);
        else if (!memcmp(f + pos + 4, "IEND", 4)) { iend = true; pos += 12 + len; break; }
        else return false;
        pos += 12 + len;
    }
    return ihdr && iend && pos == n;
}

static bool same_pixels(const std::vector<uint8_t>& idat, const uint8_t* px, int w, int h, int c, size_t stride) {
    const size_t rb = (size_t)w * c + 1;
    std::vector<uint8_t> raw(rb * h + 16);
    uLongf rn = raw.size();
    if (uncompress(raw.data(), &rn, idat.data(), idat.size()) != Z_OK || rn != rb * h) return false;   // checks the Adler-32 too
    for (int y = 0; y < h; ++y) {
        const uint8_t* r = &raw[y * rb];
        if (r[0] != 1) return false;
        for (size_t k = 0; k < (size_t)w * c; ++k) {
            const uint8_t v = (uint8_t)(r[1 + k] + (k >= (size_t)c ? px[y * stride + k - c] : 0));
            if (v != px[y * stride + k]) return false;
        }
    }
    return true;
}

#define CHECK(cond) do { if (!(cond)) { fprintf(stderr, "FAIL %s:%d %s (case %d: %dx%dx%d kind %d)\n", __FILE__, __LINE__, #cond, ncase, w, h, c, kind); return 1; } } while (0)

int main(int argc, char** argv) {
    const int cases = argc > 1 ? atoi(argv[1]) : 200;
    int ncase = 0, w = 0, h = 0, c = 0, kind = 0;
    for (; ncase < cases; ++ncase) {
        kind = ncase % 6;
        c = 3 + (ncase / 6) % 2;
        w = ncase < 12 ? 1 + ncase : 1 + (int)(rnd() % (ncase % 23 == 0 ? 3000 : 300));
        h = ncase < 12 ? 1 + ncase % 3 : 1 + (int)(rnd() % (ncase % 29 == 0 ? 900 : 200));
        std::vector<uint8_t> px;
        const size_t pad = rnd() % 5;                                  // rows further apart than they are long
        const size_t stride = (size_t)w * c + pad;
        {
            std::vector<uint8_t> t;
            fill(t, w, h, c, kind);
            px.assign(stride * h, 0xAB);
            for (int y = 0; y < h; ++y) memcpy(&px[y * stride], &t[(size_t)y * w * c], (size_t)w * c);
        }
        // whole file, exact capacity
        const size_t cap = s2sr_png_bound(w, h, c);
        uint8_t* out = (uint8_t*)malloc(cap);
        size_t n = 0;
        CHECK(s2sr_png_encode(px.data(), w, h, c, stride, out, cap, &n) == S2SR_OK && n <= cap);
        std::vector<uint8_t> idat;
        CHECK(parse(out, n, w, h, c, idat));
        CHECK(same_pixels(idat, px.data(), w, h, c, stride));
        // too little room: an error, never an overrun
        {
            const size_t small = rnd() % n;
            uint8_t* o2 = (uint8_t*)malloc(small ? small : 1);
            size_t n2 = 0;
            CHECK(s2sr_png_encode(px.data(), w, h, c, stride, o2, small, &n2) == S2SR_E_CAPACITY);
            free(o2);
        }
        free(out);
        // the same image in bands: chunks concatenate into one stream
        {
            const int br = 1 + (int)(rnd() % (h < 40 ? h : 40));
            std::vector<uint8_t> stream;
            uint32_t adler = 1;
            for (int y0 = 0; y0 < h; y0 += br) {
                const int rows = h - y0 < br ? h - y0 : br;
                const size_t bc = s2sr_png_bound(w, rows, c);
                uint8_t* bo = (uint8_t*)malloc(bc);
                size_t bn = 0, rawn = 0;
                uint32_t a = 0;
                CHECK(s2sr_png_idat_band(&px[(size_t)y0 * stride], w, rows, c, stride, y0 == 0, y0 + rows == h, bo, bc, &bn, &a, &rawn) == S2SR_OK && bn <= bc);
                CHECK(bn >= 12 && be32(bo) == bn - 12 && !memcmp(bo + 4, "IDAT", 4));
                CHECK((uint32_t)crc32(crc32(0, Z_NULL, 0), bo + 4, bn - 8) == be32(bo + bn - 4));
                stream.insert(stream.end(), bo + 8, bo + bn - 4);
                adler = (uint32_t)adler32_combine(adler, a, (z_off_t)rawn);
                CHECK(rawn == ((size_t)w * c + 1) * rows);
                free(bo);
            }
            const uint8_t ad[4] = {(uint8_t)(adler >> 24), (uint8_t)(adler >> 16), (uint8_t)(adler >> 8), (uint8_t)adler};
            stream.insert(stream.end(), ad, ad + 4);
            CHECK(same_pixels(stream, px.data(), w, h, c, stride));
        }
    }
    // argument checks
    uint8_t b[64];
    size_t k = 0;
    w = h = 1; c = 3; kind = -1;
    CHECK(s2sr_png_encode(nullptr, 1, 1, 3, 3, b, 64, &k) == S2SR_E_INVALID);
    CHECK(s2sr_png_encode(b, 1, 1, 2, 2, b + 8, 56, &k) == S2SR_E_INVALID);
    CHECK(s2sr_png_encode(b, 4, 1, 3, 5, b + 16, 48, &k) == S2SR_E_INVALID);
    CHECK(s2sr_png_bound(0, 1, 3) == 0);
    printf("ok %d cases\n", ncase);
    return 0;
}
