// Host-only PNG encoder behind the C ABI (no device code): 8-bit RGB / RGBA, filter Sub on every row, deflate with run-length
// matches only (distance 1) and dynamic Huffman blocks -- the settings cv2.imwrite uses when the reference calls it bare
// (reference server/app/wow_sr.py:156,163: Sub, Z_BEST_SPEED, Z_RLE) and what a z10-18 tile pyramid (12.8k RGBA tiles, 3.3 GB of
// pixels for one 4096x4096 SR raster; reference server/app/tiling.py:138-186 hands that to gdal2tiles) spends its time in.
// zlib's deflate_rle does ~110 MB/s per thread on that data; this file does the same tokenisation in one pass over the filtered
// rows, one Huffman build per <= 64k tokens and a 64-bit bit writer.  Any deflate stream decodes to the same pixels; the bytes
// of the file differ from zlib's (block boundaries, tree tie-breaks) and nothing pins them.
#include <errno.h>
#include <fcntl.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <immintrin.h>
#include <sys/stat.h>
#include <sys/uio.h>
#include <unistd.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../include/s2sr.h"
#include "png_internal.h"

namespace {

// ---- checksums ------------------------------------------------------------------------------
struct CrcTables {
    uint32_t t[8][256];
    CrcTables() {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            t[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; ++i)
            for (int s = 1; s < 8; ++s) t[s][i] = (t[s - 1][i] >> 8) ^ t[0][t[s - 1][i] & 0xFF];
    }
};
const CrcTables& crc_tables() {
    static const CrcTables T;
    return T;
}
}  // namespace
namespace {
uint32_t crc32_tables(uint32_t c /* running value, inverted */, const uint8_t* p, size_t n) {      // slice-by-8
    const CrcTables& T = crc_tables();
    while (n >= 8) {
        uint32_t a, b;
        memcpy(&a, p, 4);
        memcpy(&b, p + 4, 4);
        a ^= c;
        c = T.t[7][a & 0xFF] ^ T.t[6][(a >> 8) & 0xFF] ^ T.t[5][(a >> 16) & 0xFF] ^ T.t[4][a >> 24] ^
            T.t[3][b & 0xFF] ^ T.t[2][(b >> 8) & 0xFF] ^ T.t[1][(b >> 16) & 0xFF] ^ T.t[0][b >> 24];
        p += 8;
        n -= 8;
    }
    while (n--) c = T.t[0][(c ^ *p++) & 0xFF] ^ (c >> 8);
    return c;
}

// CRC-32 by carry-less multiplication (Gopal et al., "Fast CRC Computation for Generic Polynomials Using PCLMULQDQ", the
// bit-reflected form zlib's x86 builds use): four 128-bit lanes are folded over 64 bytes at a time with x^(512+-32) mod P, then into
// one lane with x^(128+-32) mod P, then 128 -> 64 -> 32 bits with a Barrett reduction.  n >= 64 and a multiple of 16; ~10x the table
// walk, which was 40 % of a tile file's host time in the pyramid's PNG stage (12.8k files of ~30 KB per pyramid).
__attribute__((target("pclmul,sse4.1"))) uint32_t crc32_clmul(uint32_t c /* running value, inverted */, const uint8_t* p, size_t n) {
    const __m128i k1k2 = _mm_set_epi64x(0x01c6e41596, 0x0154442bd4);      // x^(4*128+32), x^(4*128-32) mod P (reflected)
    const __m128i k3k4 = _mm_set_epi64x(0x00ccaa009e, 0x01751997d0);      // x^(128+32), x^(128-32) mod P
    const __m128i k5 = _mm_set_epi64x(0, 0x0163cd6124);                   // x^64 mod P
    const __m128i poly = _mm_set_epi64x(0x01f7011641, 0x01db710641);      // mu = floor(x^64 / P), P
    __m128i x1 = _mm_loadu_si128((const __m128i*)(p + 0)), x2 = _mm_loadu_si128((const __m128i*)(p + 16));
    __m128i x3 = _mm_loadu_si128((const __m128i*)(p + 32)), x4 = _mm_loadu_si128((const __m128i*)(p + 48));
    x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)c));
    p += 64;
    n -= 64;
    while (n >= 64) {
        const __m128i l1 = _mm_clmulepi64_si128(x1, k1k2, 0x00), l2 = _mm_clmulepi64_si128(x2, k1k2, 0x00);
        const __m128i l3 = _mm_clmulepi64_si128(x3, k1k2, 0x00), l4 = _mm_clmulepi64_si128(x4, k1k2, 0x00);
        x1 = _mm_clmulepi64_si128(x1, k1k2, 0x11);
        x2 = _mm_clmulepi64_si128(x2, k1k2, 0x11);
        x3 = _mm_clmulepi64_si128(x3, k1k2, 0x11);
        x4 = _mm_clmulepi64_si128(x4, k1k2, 0x11);
        x1 = _mm_xor_si128(_mm_xor_si128(x1, l1), _mm_loadu_si128((const __m128i*)(p + 0)));
        x2 = _mm_xor_si128(_mm_xor_si128(x2, l2), _mm_loadu_si128((const __m128i*)(p + 16)));
        x3 = _mm_xor_si128(_mm_xor_si128(x3, l3), _mm_loadu_si128((const __m128i*)(p + 32)));
        x4 = _mm_xor_si128(_mm_xor_si128(x4, l4), _mm_loadu_si128((const __m128i*)(p + 48)));
        p += 64;
        n -= 64;
    }
    auto fold1 = [&](__m128i acc, __m128i next) __attribute__((target("pclmul,sse4.1"))) {
        const __m128i lo = _mm_clmulepi64_si128(acc, k3k4, 0x00);
        return _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(acc, k3k4, 0x11), next), lo);
    };
    x1 = fold1(x1, x2);
    x1 = fold1(x1, x3);
    x1 = fold1(x1, x4);
    while (n >= 16) {
        x1 = fold1(x1, _mm_loadu_si128((const __m128i*)p));
        p += 16;
        n -= 16;
    }
    // 128 -> 64 bits
    const __m128i mask32 = _mm_setr_epi32(~0, 0, ~0, 0);
    __m128i t = _mm_clmulepi64_si128(x1, k3k4, 0x10);
    x1 = _mm_xor_si128(_mm_srli_si128(x1, 8), t);
    t = _mm_srli_si128(x1, 4);
    x1 = _mm_xor_si128(_mm_clmulepi64_si128(_mm_and_si128(x1, mask32), k5, 0x00), t);
    // Barrett reduction to 32 bits
    t = _mm_and_si128(_mm_clmulepi64_si128(_mm_and_si128(x1, mask32), poly, 0x10), mask32);
    x1 = _mm_xor_si128(x1, _mm_clmulepi64_si128(t, poly, 0x00));
    return (uint32_t)_mm_extract_epi32(x1, 1);
}

bool have_clmul() {
    static const bool ok = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1") && !getenv("S2SR_CRC_TABLES");
    return ok;
}
}  // namespace

uint32_t s2sr::png::crc32_update(uint32_t crc, const uint8_t* p, size_t n) {      // crc is the running value (not inverted)
    uint32_t c = ~crc;
    if (n >= 64 && have_clmul()) {
        const size_t body = n & ~(size_t)15;
        c = crc32_clmul(c, p, body);
        p += body;
        n -= body;
    }
    return ~crc32_tables(c, p, n);
}
uint32_t s2sr::png::adler32_update(uint32_t adler, const uint8_t* p, size_t n) {
    uint32_t a = adler & 0xFFFF;
    uint64_t b = adler >> 16;
    while (n) {
        size_t k = n < 5552 ? n : 5552;              // a stays below 2^32 over this many bytes
        n -= k;
        for (; k >= 32; k -= 32, p += 32) {          // 32 bytes at once: b += 32 a + sum (32 - j) p[j]; constant weights vectorise
            uint32_t s1 = 0, s2 = 0;
            for (int j = 0; j < 32; ++j) { s1 += p[j]; s2 += (uint32_t)(32 - j) * p[j]; }
            b += 32ull * a + s2;
            a += s1;
        }
        for (; k; --k) { a += *p++; b += a; }
        a %= 65521; b %= 65521;
    }
    return ((uint32_t)b << 16) | a;
}

namespace {
using namespace s2sr::png;

inline uint64_t load64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }

inline void put32be(uint8_t* p, uint32_t v) { p[0] = (uint8_t)(v >> 24); p[1] = (uint8_t)(v >> 16); p[2] = (uint8_t)(v >> 8); p[3] = (uint8_t)v; }

// ---- Huffman --------------------------------------------------------------------------------
// Code lengths (<= maxbits) for n symbols from their frequencies; unused symbols get 0.  Two-queue construction on the sorted
// used symbols, then the classic repair of the length histogram when the tree is deeper than maxbits.
void huffman_lengths(const uint32_t* freq, int n, int maxbits, uint8_t* len) {
    struct Leaf { uint32_t f; int s; };
    Leaf leaves[288];
    int m = 0;
    for (int i = 0; i < n; ++i) {
        len[i] = 0;
        if (freq[i]) leaves[m++] = Leaf{freq[i], i};
    }
    if (m == 0) return;
    if (m == 1) { len[leaves[0].s] = 1; return; }
    std::sort(leaves, leaves + m, [](const Leaf& a, const Leaf& b) { return a.f < b.f || (a.f == b.f && a.s < b.s); });
    uint64_t w[576];
    int parent[576];
    for (int i = 0; i < m; ++i) w[i] = leaves[i].f;
    int li = 0, ni = m, k = m;                       // next leaf, next unmerged internal node, next internal node to create
    for (; k < 2 * m - 1; ++k) {
        int pick[2];
        for (int t = 0; t < 2; ++t) {
            if (li < m && (ni >= k || w[li] <= w[ni])) pick[t] = li++;
            else pick[t] = ni++;
        }
        w[k] = w[pick[0]] + w[pick[1]];
        parent[pick[0]] = parent[pick[1]] = k;
    }
    int depth[576];
    depth[2 * m - 2] = 0;
    int count[64] = {0};
    for (int i = 2 * m - 3; i >= 0; --i) {
        depth[i] = depth[parent[i]] + 1;
        if (i < m) ++count[depth[i] < 63 ? depth[i] : 63];
    }
    // fold the too-deep leaves into maxbits and pay for them by pushing shallower leaves down (Kraft sum back to 1)
    bool deep = false;
    for (int d = maxbits + 1; d < 64; ++d) if (count[d]) { count[maxbits] += count[d]; count[d] = 0; deep = true; }
    if (deep) {
        uint64_t total = 0;
        for (int d = 1; d <= maxbits; ++d) total += (uint64_t)count[d] << (maxbits - d);
        while (total > ((uint64_t)1 << maxbits)) {
            bool moved = false;
            for (int d = maxbits - 1; d > 0 && !moved; --d)
                if (count[d]) { --count[d]; count[d + 1] += 2; moved = true; }
            if (!moved) break;                       // cannot happen for n <= 2^maxbits symbols
            --count[maxbits];
            --total;
        }
    }
    // the rarest symbols take the longest codes
    int i = 0;
    for (int d = maxbits; d >= 1; --d)
        for (int c = count[d]; c > 0; --c) len[leaves[i++].s] = (uint8_t)d;
}

// canonical codes, bit-reversed for the LSB-first bit writer
void huffman_codes(const uint8_t* len, int n, uint16_t* code) {
    int bl_count[16] = {0};
    for (int i = 0; i < n; ++i) ++bl_count[len[i]];
    bl_count[0] = 0;
    int next[16];
    int c = 0;
    for (int b = 1; b < 16; ++b) { c = (c + bl_count[b - 1]) << 1; next[b] = c; }
    for (int i = 0; i < n; ++i) {
        if (!len[i]) { code[i] = 0; continue; }
        int v = next[len[i]]++, r = 0;
        for (int b = 0; b < len[i]; ++b) { r = (r << 1) | (v & 1); v >>= 1; }
        code[i] = (uint16_t)r;
    }
}

struct BitWriter {
    uint8_t* p;
    uint8_t* end;
    uint64_t acc = 0;
    int n = 0;
    bool ok = true;
    inline void put(uint32_t bits, int nbits) {      // nbits <= 32
        acc |= (uint64_t)bits << n;
        n += nbits;
        if (n >= 32) {
            if (end - p < 4) { ok = false; n = 0; acc = 0; return; }
            const uint32_t v = (uint32_t)acc;
            memcpy(p, &v, 4);                        // little endian host
            p += 4;
            acc >>= 32;
            n -= 32;
        }
    }
    // Fast path for the token stream of a block whose size was checked against the buffer beforehand (8 bytes of slack): whole
    // bytes leave the accumulator at every step (an unaligned 8-byte store, the pointer moves by the bytes completed), so at
    // most 7 bits are pending and a step may add up to 56.
    inline void fast_begin() {
        while (n >= 8) { *p++ = (uint8_t)acc; acc >>= 8; n -= 8; }
    }
    inline void fast_put(uint64_t bits, int nbits) {
        acc |= bits << n;
        n += nbits;
        memcpy(p, &acc, 8);                          // little endian host
        p += n >> 3;
        acc >>= n & ~7;
        n &= 7;
    }
    inline void fast_end() {}
    void align() {                                   // to a byte boundary
        while (n > 0) {
            if (p >= end) { ok = false; n = 0; return; }
            *p++ = (uint8_t)acc;
            acc >>= 8;
            n -= 8;
        }
        n = 0;
        acc = 0;
    }
};

const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t kClOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

struct LenSym {
    uint16_t sym[256];      // match length - 3 -> length symbol 257..285
    LenSym() {
        int s = 0;
        for (int l = 3; l <= 258; ++l) {
            while (s < 28 && l >= kLenBase[s + 1]) ++s;
            sym[l - 3] = (uint16_t)(257 + s);
        }
    }
};
const LenSym& len_syms() {
    static const LenSym L;
    return L;
}

}  // namespace

void s2sr::png::build_block_code(const uint32_t* tok_freq, bool final_block, BlockCode* bc) {
    const LenSym& LS = len_syms();
    uint32_t freq[286] = {0};
    for (int t = 0; t < 256; ++t) freq[t] = tok_freq[t];
    for (int t = 256; t < 512; ++t) freq[LS.sym[t - 256]] += tok_freq[t];
    freq[256] = 1;
    uint8_t ll_len[286];
    uint16_t ll_code[286];
    huffman_lengths(freq, 286, 15, ll_len);
    int used = 0;
    for (int i = 0; i < 286; ++i) used += ll_len[i] != 0;
    if (used < 2) ll_len[ll_len[0] ? 1 : 0] = 1;     // a complete code needs two symbols (an empty block: EOB only)
    huffman_codes(ll_len, 286, ll_code);
    int hlit = 286;
    while (hlit > 257 && !ll_len[hlit - 1]) --hlit;
    // the code lengths (literal/length alphabet, then ONE distance code of one bit), run-length coded over the 19-symbol alphabet
    uint8_t seq[287];
    memcpy(seq, ll_len, hlit);
    seq[hlit] = 1;
    const int nseq = hlit + 1;
    uint8_t cl_sym[287], cl_extra[287];
    int ncl = 0;
    uint32_t cl_freq[19] = {0};
    for (int i = 0; i < nseq;) {
        int run = 1;
        while (i + run < nseq && seq[i + run] == seq[i]) ++run;
        const int v = seq[i];
        int left = run;
        if (v == 0) {
            while (left >= 11) { const int r = left < 138 ? left : 138; cl_sym[ncl] = 18; cl_extra[ncl++] = (uint8_t)(r - 11); left -= r; }
            if (left >= 3) { cl_sym[ncl] = 17; cl_extra[ncl++] = (uint8_t)(left - 3); left = 0; }
            while (left-- > 0) { cl_sym[ncl] = 0; cl_extra[ncl++] = 0; }
        } else {
            cl_sym[ncl] = (uint8_t)v; cl_extra[ncl++] = 0; --left;
            while (left >= 3) { const int r = left < 6 ? left : 6; cl_sym[ncl] = 16; cl_extra[ncl++] = (uint8_t)(r - 3); left -= r; }
            while (left-- > 0) { cl_sym[ncl] = (uint8_t)v; cl_extra[ncl++] = 0; }
        }
        i += run;
    }
    for (int i = 0; i < ncl; ++i) ++cl_freq[cl_sym[i]];
    uint8_t cl_len[19];
    uint16_t cl_code[19];
    huffman_lengths(cl_freq, 19, 7, cl_len);
    {
        int u = 0;
        for (int i = 0; i < 19; ++i) u += cl_len[i] != 0;
        if (u < 2) cl_len[cl_len[0] ? 1 : 0] = 1;
    }
    huffman_codes(cl_len, 19, cl_code);
    int hclen = 19;
    while (hclen > 4 && !cl_len[kClOrder[hclen - 1]]) --hclen;
    // the block header into bc->header (at most 17 + 57 + 287 * 14 bits)
    memset(bc->header, 0, sizeof bc->header);
    BitWriter hw;
    hw.p = bc->header;
    hw.end = bc->header + sizeof bc->header;
    hw.put((final_block ? 1 : 0) | (2 << 1), 3);
    hw.put(hlit - 257, 5);
    hw.put(0, 5);                                    // HDIST - 1
    hw.put(hclen - 4, 4);
    for (int i = 0; i < hclen; ++i) hw.put(cl_len[kClOrder[i]], 3);
    for (int i = 0; i < ncl; ++i) {
        const int s = cl_sym[i];
        hw.put(cl_code[s], cl_len[s]);
        if (s == 16) hw.put(cl_extra[i], 2);
        else if (s == 17) hw.put(cl_extra[i], 3);
        else if (s == 18) hw.put(cl_extra[i], 7);
    }
    bc->header_bits = (uint32_t)((hw.p - bc->header) * 8 + hw.n);
    hw.align();
    // token -> (bits, count): a literal's code; a match's length code | extra bits | the 1-bit distance code 0
    for (int i = 0; i < 256; ++i) bc->tb[i] = ll_code[i] | ((uint32_t)ll_len[i] << 24);
    for (int l = 0; l < 256; ++l) {
        const int sy = LS.sym[l];
        if (!ll_len[sy]) { bc->tb[256 + l] = 0; continue; }
        const int eb = kLenExtra[sy - 257];
        bc->tb[256 + l] = (ll_code[sy] | ((uint32_t)((l + 3) - kLenBase[sy - 257]) << ll_len[sy])) | ((uint32_t)(ll_len[sy] + eb + 1) << 24);
    }
    bc->eob = ll_code[256] | ((uint32_t)ll_len[256] << 24);
    uint64_t bits = ll_len[256];
    for (int t = 0; t < 512; ++t) bits += (uint64_t)tok_freq[t] * (bc->tb[t] >> 24);
    bc->body_bits = bits;
}

namespace {

// One deflate block: tokens [t0, t1) covering raw[r0, r1).
void emit_block(BitWriter& bw, const uint16_t* tok, size_t t0, size_t t1, const uint8_t* raw, size_t r0, size_t r1, bool final_block) {
    // token histogram over 512 slots (literals, then match lengths), four tables so that a repeated token does not wait for
    // its own previous increment (filtered imagery is mostly 0x00 / 0x01 / 0xFF)
    uint32_t h4[4][512];
    memset(h4, 0, sizeof h4);
    size_t hi = t0;
    for (; hi + 4 <= t1; hi += 4) { ++h4[0][tok[hi]]; ++h4[1][tok[hi + 1]]; ++h4[2][tok[hi + 2]]; ++h4[3][tok[hi + 3]]; }
    for (; hi < t1; ++hi) ++h4[0][tok[hi]];
    for (int t = 0; t < 512; ++t) h4[0][t] += h4[1][t] + h4[2][t] + h4[3][t];
    BlockCode bc;
    build_block_code(h4[0], final_block, &bc);
    const uint64_t bits = bc.header_bits + bc.body_bits;
    const size_t nraw = r1 - r0;
    const uint64_t stored_bits = 8 * (uint64_t)nraw + 40 * ((nraw + 65534) / 65535 + (nraw == 0)) + 7;
    if (bits >= stored_bits) {
        size_t at = r0;
        do {
            const size_t k = r1 - at < 65535 ? r1 - at : 65535;
            bw.put((final_block && at + k == r1) ? 1 : 0, 3);          // BFINAL, BTYPE = 00
            bw.align();
            if (!bw.ok || (size_t)(bw.end - bw.p) < 4 + k) { bw.ok = false; return; }
            bw.p[0] = (uint8_t)k; bw.p[1] = (uint8_t)(k >> 8); bw.p[2] = (uint8_t)~k; bw.p[3] = (uint8_t)(~k >> 8);
            memcpy(bw.p + 4, raw + at, k);
            bw.p += 4 + k;
            at += k;
        } while (at < r1);
        return;
    }
    if ((uint64_t)(bw.end - bw.p) < bits / 8 + 32) { bw.ok = false; return; }
    for (uint32_t k = 0; k < bc.header_bits; k += 8) bw.put(bc.header[k >> 3], bc.header_bits - k < 8 ? (int)(bc.header_bits - k) : 8);
    // two tokens per step through the byte-granular writer (<= 42 bits on top of <= 7 pending)
    const uint32_t* tb = bc.tb;
    bw.fast_begin();
    size_t i = t0;
    for (; i + 2 <= t1; i += 2) {
        const uint32_t e0 = tb[tok[i]], e1 = tb[tok[i + 1]];
        const int n0 = e0 >> 24;
        bw.fast_put((uint64_t)(e0 & 0xFFFFFF) | ((uint64_t)(e1 & 0xFFFFFF) << n0), n0 + (int)(e1 >> 24));
    }
    if (i < t1) bw.fast_put(tb[tok[i]] & 0xFFFFFF, (int)(tb[tok[i]] >> 24));
    bw.fast_end();
    bw.put(bc.eob & 0xFFFFFF, (int)(bc.eob >> 24));
}

// Sub-filter `rows` rows of c-byte pixels into raw (rows x (1 + w*c)), tokenise, emit deflate blocks.  `finish`: the last block
// carries BFINAL; otherwise the piece ends on an empty stored block (a sync flush: byte aligned, the next piece can follow).
// Returns bytes written or (size_t)-1 when out is too small; *adler = Adler-32 of the filtered bytes (running value in / out).
size_t deflate_rows(const uint8_t* px, int w, int rows, int c, size_t stride, bool finish, uint8_t* out, size_t cap, uint32_t* adler,
                    uint8_t* raw, uint16_t* tok) {
    const size_t rb = (size_t)w * c + 1, n = rb * rows;
    for (int y = 0; y < rows; ++y) {
        const uint8_t* s = px + (size_t)y * stride;
        uint8_t* d = raw + (size_t)y * rb;
        d[0] = 1;                                    // filter type Sub, bpp = c
        for (int k = 0; k < c; ++k) d[1 + k] = s[k];
        const size_t m = (size_t)w * c;
        for (size_t k = c; k < m; ++k) d[1 + k] = (uint8_t)(s[k] - s[k - c]);
    }
    *adler = adler32_update(*adler, raw, n);
    BitWriter bw;
    bw.p = out;
    bw.end = out + cap;
    const size_t kBlockTokens = 65536;
    size_t nt = 0, r0 = 0, i = 0;
    auto flush = [&](bool last) {
        emit_block(bw, tok, 0, nt, raw, r0, i, last && finish);
        nt = 0;
        r0 = i;
    };
    while (i < n) {
        if (nt >= kBlockTokens - 8) {                // room for the 6 literals + 1 match a step can add
            flush(false);
            if (!bw.ok) return (size_t)-1;
        }
        if (i == 0 || i + 8 > n) {                   // the first byte has no predecessor; the tail goes byte by byte
            const uint8_t v = raw[i];
            if (i > 0 && raw[i - 1] == v && i + 2 < n && raw[i + 1] == v && raw[i + 2] == v) {
                size_t run = 3;
                const size_t lim = n - i < 258 ? n - i : 258;
                while (run < lim && raw[i + run] == v) ++run;
                tok[nt++] = (uint16_t)(256 + run - 3);
                i += run;
            } else {
                tok[nt++] = v;
                ++i;
            }
            continue;
        }
        // eight positions at once: byte j of x is zero where raw[i + j] == raw[i + j - 1]; a match (distance 1, length >= 3)
        // starts at the first j with three zero bytes in a row.  Starts 0..5 are decided here, 6 and 7 by the next step.
        const uint64_t cur = load64(raw + i);
        const uint64_t x = cur ^ load64(raw + i - 1);
        const uint64_t k7 = 0x7F7F7F7F7F7F7F7Full;
        const uint64_t z = ~(((x & k7) + k7) | x | k7);                      // 0x80 in every zero byte of x, exactly
        const uint64_t t3 = z & (z >> 8) & (z >> 16) & 0x0000808080808080ull;
        const int first = t3 ? (__builtin_ctzll(t3) >> 3) : 6;
        for (int j = 0; j < 6; ++j) tok[nt + j] = (uint8_t)(cur >> (8 * j));      // all six; the ones past `first` are overwritten
        nt += first;
        i += first;
        if (!t3) continue;
        const uint8_t v = raw[i];
        const uint64_t vv = 0x0101010101010101ull * v;
        const size_t lim = n - i < 258 ? n - i : 258;
        size_t run = 3;
        bool open = true;
        while (run + 8 <= lim) {                     // raw has 8 bytes of slack behind n
            const uint64_t y = load64(raw + i + run) ^ vv;
            if (y) { run += __builtin_ctzll(y) >> 3; open = false; break; }
            run += 8;
        }
        if (open) while (run < lim && raw[i + run] == v) ++run;
        tok[nt++] = (uint16_t)(256 + run - 3);
        i += run;
    }
    flush(true);
    if (!bw.ok) return (size_t)-1;
    if (!finish) {                                   // sync flush: an empty stored block
        bw.put(0, 3);
        bw.align();
        if (!bw.ok || bw.end - bw.p < 4) return (size_t)-1;
        bw.p[0] = 0; bw.p[1] = 0; bw.p[2] = 0xFF; bw.p[3] = 0xFF;
        bw.p += 4;
    } else {
        bw.align();
        if (!bw.ok) return (size_t)-1;
    }
    return (size_t)(bw.p - out);
}

size_t chunk(uint8_t* out, const char* kind, const uint8_t* data, size_t n) {      // data may already sit at out + 8
    put32be(out, (uint32_t)n);
    memcpy(out + 4, kind, 4);
    if (data != out + 8 && n) memmove(out + 8, data, n);
    put32be(out + 8 + n, crc32_update(0, out + 4, n + 4));
    return n + 12;
}

// filtered rows + token buffer of the calling thread, kept between calls: a pyramid encodes 12.8k tiles on a few dozen threads,
// and 400 KB of malloc / free per tile is three mmap / munmap pairs under the process's address-space lock
struct Scratch {
    uint8_t* raw = nullptr;
    size_t raw_cap = 0;
    uint16_t* tok = nullptr;
    ~Scratch() { free(raw); free(tok); }
    bool reserve(size_t n) {
        if (!tok) tok = (uint16_t*)malloc(65536 * sizeof(uint16_t));
        if (raw_cap < n + 8) {
            free(raw);
            raw_cap = 0;
            raw = (uint8_t*)malloc(n + 8);
            if (raw) raw_cap = n + 8;
        }
        if (raw && raw_cap > ((size_t)8 << 20) && n + 8 < raw_cap / 4) {   // a one-off big band does not pin its buffer forever
            uint8_t* r = (uint8_t*)realloc(raw, n + 8);
            if (r) { raw = r; raw_cap = n + 8; }
        }
        return raw && tok;
    }
};
Scratch& scratch() {
    static thread_local Scratch s;
    return s;
}

}  // namespace

extern "C" size_t s2sr_png_bound(int32_t width, int32_t rows, int32_t channels) {
    if (width <= 0 || rows <= 0 || channels <= 0) return 0;
    const size_t n = ((size_t)width * channels + 1) * rows;
    return n + 5 * (n / 65535 + 2) + 600 * (n / 65536 + 2) + 128;     // stored worst case; a tree header per block; framing
}

extern "C" int s2sr_png_encode(const uint8_t* px, int32_t width, int32_t height, int32_t channels, size_t row_stride, uint8_t* out,
                               size_t cap, size_t* out_n) {
    if (!px || !out || !out_n || width <= 0 || height <= 0 || (channels != 3 && channels != 4) || row_stride < (size_t)width * channels)
        return S2SR_E_INVALID;
    const size_t rb = (size_t)width * channels + 1, n = rb * height;
    if (n >= ((size_t)1 << 31) - 65536) return S2SR_E_INVALID;         // one IDAT chunk; bigger images go through the band call
    if (cap < 8 + 25 + 12 + 6 + 12) return S2SR_E_CAPACITY;
    Scratch& sc = scratch();
    if (!sc.reserve(n)) return S2SR_E_CAPACITY;
    uint8_t* raw = sc.raw;
    uint16_t* tok = sc.tok;
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
    memcpy(out, sig, 8);
    uint8_t ihdr[13];
    put32be(ihdr, (uint32_t)width);
    put32be(ihdr + 4, (uint32_t)height);
    ihdr[8] = 8; ihdr[9] = channels == 3 ? 2 : 6; ihdr[10] = ihdr[11] = ihdr[12] = 0;
    size_t pos = 8 + chunk(out + 8, "IHDR", ihdr, 13);
    uint8_t* idat = out + pos + 8;
    idat[0] = 0x78; idat[1] = 0x01;
    uint32_t adler = 1;
    const size_t room = cap - pos - 8 - 2;
    const size_t dn = room > 32 ? deflate_rows(px, width, height, channels, row_stride, true, idat + 2, room - 32, &adler, raw, tok) : (size_t)-1;
    if (dn == (size_t)-1) return S2SR_E_CAPACITY;
    put32be(idat + 2 + dn, adler);
    pos += chunk(out + pos, "IDAT", idat, 2 + dn + 4);
    pos += chunk(out + pos, "IEND", nullptr, 0);
    *out_n = pos;
    return S2SR_OK;
}

extern "C" int s2sr_png_idat_band(const uint8_t* px, int32_t width, int32_t rows, int32_t channels, size_t row_stride, int32_t first,
                                  int32_t last, uint8_t* out, size_t cap, size_t* out_n, uint32_t* adler, size_t* raw_n) {
    if (!px || !out || !out_n || !adler || !raw_n || width <= 0 || rows <= 0 || (channels != 3 && channels != 4) ||
        row_stride < (size_t)width * channels)
        return S2SR_E_INVALID;
    const size_t rb = (size_t)width * channels + 1, n = rb * rows;
    if (n >= ((size_t)1 << 31) - 65536) return S2SR_E_INVALID;
    if (cap < 64) return S2SR_E_CAPACITY;
    Scratch& sc = scratch();
    if (!sc.reserve(n)) return S2SR_E_CAPACITY;
    uint8_t* raw = sc.raw;
    uint16_t* tok = sc.tok;
    uint8_t* idat = out + 8;
    size_t head = 0;
    if (first) { idat[0] = 0x78; idat[1] = 0x01; head = 2; }
    uint32_t a = 1;
    const size_t dn = deflate_rows(px, width, rows, channels, row_stride, last != 0, idat + head, cap - 8 - head - 16, &a, raw, tok);
    if (dn == (size_t)-1) return S2SR_E_CAPACITY;
    *adler = a;
    *raw_n = n;
    *out_n = chunk(out, "IDAT", idat, head + dn);
    return S2SR_OK;
}

namespace {
int open_for_write(const char* path) {
    int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0644);
    if (fd < 0 && errno == ENOENT) {                 // z/x/ does not exist yet: make the missing directories, once
        std::string p(path);
        for (size_t k = 1; k < p.size(); ++k)
            if (p[k] == '/') {
                p[k] = 0;
                if (mkdir(p.c_str(), 0755) != 0 && errno != EEXIST) return -1;
                p[k] = '/';
            }
        fd = open(path, O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0644);
    }
    return fd;
}
}  // namespace

bool s2sr::png::write_file(const char* path, const uint8_t* data, size_t n) {
    const Piece one = {data, n};
    return write_file_pieces(path, &one, 1);
}

bool s2sr::png::write_file_pieces(const char* path, const Piece* pieces, int count) {
    if (count < 1 || count > 8) return false;
    const int fd = open_for_write(path);
    if (fd < 0) return false;
    struct iovec iov[8];
    int n = 0;
    for (int i = 0; i < count; ++i)
        if (pieces[i].n) { iov[n].iov_base = (void*)pieces[i].p; iov[n].iov_len = pieces[i].n; ++n; }
    int at = 0;
    while (at < n) {
        const ssize_t w = writev(fd, iov + at, n - at);
        if (w < 0) {
            if (errno == EINTR) continue;
            close(fd);
            return false;
        }
        size_t left = (size_t)w;                     // a short write: step over what went out
        while (at < n && left >= iov[at].iov_len) left -= iov[at++].iov_len;
        if (at < n && left) { iov[at].iov_base = (char*)iov[at].iov_base + left; iov[at].iov_len -= left; }
    }
    return close(fd) == 0;
}

extern "C" int s2sr_png_write_tiles(const uint8_t* tiles, int32_t count, int32_t size, int32_t channels, size_t tile_stride,
                                    const char* const* paths, int32_t skip_transparent, int32_t* written) {
    if (!tiles || !paths || count < 0 || size <= 0 || (channels != 3 && channels != 4) || tile_stride < (size_t)size * size * channels)
        return S2SR_E_INVALID;
    const size_t cap = s2sr_png_bound(size, size, channels);
    std::vector<uint8_t> out(cap);
    for (int32_t t = 0; t < count; ++t) {
        const uint8_t* px = tiles + (size_t)t * tile_stride;
        if (written) written[t] = 0;
        if (!paths[t]) continue;
        if (skip_transparent && channels == 4) {
            const size_t n = (size_t)size * size;
            uint32_t any = 0;
            for (size_t k = 0; k < n; ++k) any |= px[4 * k + 3];
            if (!any) continue;
        }
        size_t n = 0;
        const int rc = s2sr_png_encode(px, size, size, channels, (size_t)size * channels, out.data(), cap, &n);
        if (rc != S2SR_OK) return rc;
        if (!write_file(paths[t], out.data(), n)) return S2SR_E_IO;
        if (written) written[t] = 1;
    }
    return S2SR_OK;
}
