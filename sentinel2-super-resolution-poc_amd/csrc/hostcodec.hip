// Host-only helpers behind the C ABI (no device code): the TIFF flavour of LZW, which the file
// glue around the path needs for the LZW GeoTIFFs the reference reads and writes
// (reference server/app/wow_sr.py:59-79 reads through rasterio; :138-151 writes compress="lzw").
#include <stdint.h>
#include <string.h>

#include "../../include/s2sr.h"

extern "C" int s2sr_tiff_lzw_decode(const uint8_t* src, size_t n, uint8_t* dst, size_t cap, size_t* out_n) {
    if ((!src && n) || !dst || !out_n) return S2SR_E_INVALID;
    // TIFF 6.0 section 13: MSB-first codes of 9..12 bits, ClearCode 256, EndOfInformation 257, code width grows one code early
    // ("early change").  A table entry is (where its string first appeared in the OUTPUT, its length): string(old) + first(cur)
    // is what was written for `old` plus the byte that follows it, so decoding a code is one forward copy inside dst instead of
    // a walk down a prefix chain (r04: 150 -> ~500 MB/s per thread on image strips; reading a job's input is all LZW decode).
    static const int MAXC = 4096;
    if (cap > 0xFFFFFFFFu) cap = 0xFFFFFFFFu;                     // positions are kept in 32 bits: one strip or tile, not an image
    uint32_t pos_of[MAXC];
    uint16_t len_of[MAXC];
    int next = 258, width = 9;
    bool have_old = false;
    size_t oldpos = 0, oldlen = 0;
    size_t pos = 0, ip = 0;
    uint64_t acc = 0;
    int nacc = 0;
    for (;;) {
        while (nacc <= 56 && ip < n) { acc = (acc << 8) | src[ip++]; nacc += 8; }
        if (nacc < width) break;                                  // the stream ended without EndOfInformation
        const int code = (int)((acc >> (nacc - width)) & ((1u << width) - 1));
        nacc -= width;
        if (code == 257) break;
        if (code == 256) { next = 258; width = 9; have_old = false; continue; }
        if (pos >= cap) break;
        const size_t room = cap - pos;
        size_t len;
        if (!have_old) {                   // first code after a clear is a literal
            if (code > 255) return S2SR_E_INVALID;
            dst[pos] = (uint8_t)code;
            len = 1;
        } else {
            if (code < 256) {
                dst[pos] = (uint8_t)code;
                len = 1;
            } else {
                size_t from, l;
                bool kwkwk = false;
                if (code < next) { from = pos_of[code]; l = len_of[code]; }
                else {                     // KwKwK: the string being defined right now = string(old) + first(old)
                    if (code != next || next >= MAXC) return S2SR_E_INVALID;
                    from = oldpos; l = oldlen; kwkwk = true;
                }
                len = l + (kwkwk ? 1 : 0);
                if (len + 8 <= room) {     // 8 bytes at a time; the overshoot lands on positions not written yet
                    for (size_t k = 0; k < l; k += 8) memcpy(dst + pos + k, dst + from + k, 8);
                    if (kwkwk) dst[pos + l] = dst[oldpos];
                } else {
                    const size_t m = len < room ? len : room;
                    for (size_t k = 0; k < m; ++k) dst[pos + k] = (k < l) ? dst[from + k] : dst[oldpos];
                }
            }
            if (next < MAXC) {             // string(old) + first(cur): old's bytes and the one just written behind them
                pos_of[next] = (uint32_t)oldpos;
                len_of[next] = (uint16_t)(oldlen + 1);
                ++next;
            }
            if (next + 1 >= (1 << width) && width < 12) ++width;
        }
        oldpos = pos; oldlen = len; have_old = true;
        pos += len < room ? len : room;
        if (pos >= cap) break;
    }
    *out_n = pos;
    return S2SR_OK;
}

// TIFF LZW encoder (the counterpart of the decoder above; reference writes compress="lzw" GeoTIFFs,
// server/app/wow_sr.py:138-151).  One strip per call, so strips compress in parallel on host threads.
// Worst case output is ~1.4x the input (12-bit codes for single bytes) + a few bytes: the caller
// sizes dst as n*3/2 + 16.
// The "literal" form of a TIFF LZW stream: every byte as its own 9-bit code, a ClearCode every 250 codes so that the decoder's
// table never reaches the 10-bit switch (after a clear the first code adds nothing, each later one adds an entry: 258 + 249 = 507
// < 511).  A valid LZW stream for every decoder -- it just never uses the dictionary -- of 1.13x the input, written at memory speed.
// The SR outputs this path writes are 8-bit RGB with sensor-like texture: real LZW *expands* them (1.25 - 1.35x at ~135 MB/s per
// thread: most lookups miss, every miss costs a probe and a 9..12-bit code for one byte), so on such strips this form is both the
// smaller file and ~6x faster; s2sr_tiff_lzw_encode picks it per strip when a 16-KB sample does not compress under the dictionary coder.
static int lzw_encode_literal(const uint8_t* src, size_t n, uint8_t* dst, size_t cap, size_t* out_n) {
    const size_t need = ((n + n / 250 + 3) * 9 + 7) / 8;
    if (cap < need) return S2SR_E_CAPACITY;
    size_t pos = 0;
    uint64_t acc = 0;
    int nacc = 0;
    auto put9 = [&](uint32_t code) {
        acc = (acc << 9) | code;
        nacc += 9;
        if (nacc >= 32) {
            const uint32_t v = __builtin_bswap32((uint32_t)(acc >> (nacc - 32)));
            memcpy(dst + pos, &v, 4);
            pos += 4;
            nacc -= 32;
        }
    };
    size_t i = 0;
    while (i < n) {
        put9(256);
        const size_t end = i + 250 < n ? i + 250 : n;
        for (; i < end; ++i) put9(src[i]);
    }
    if (n == 0) put9(256);
    put9(257);
    while (nacc >= 8) { dst[pos++] = (uint8_t)(acc >> (nacc - 8)); nacc -= 8; }
    if (nacc > 0) dst[pos++] = (uint8_t)(acc << (8 - nacc));
    *out_n = pos;
    return S2SR_OK;
}

static int lzw_encode_dictionary(const uint8_t* src, size_t n, uint8_t* dst, size_t cap, size_t* out_n);

extern "C" int s2sr_tiff_lzw_encode(const uint8_t* src, size_t n, uint8_t* dst, size_t cap, size_t* out_n) {
    if ((!src && n) || !dst || !out_n) return S2SR_E_INVALID;
    // which form?  The dictionary coder on a sample from the middle of the strip: if it cannot compress the sample at all (>= 1.0x),
    // the literal form (1.13x, ~6x faster) takes the strip -- a file at most 13 % larger than the dictionary's best case there, written
    // in a sixth of the time.  Strips that do compress, and small strips, go to the dictionary coder.
    static const size_t kSample = 16384;
    if (n >= 4 * kSample) {
        uint8_t tmp[kSample * 3 / 2 + 16];
        size_t m = 0;
        if (lzw_encode_dictionary(src + (n / 2 & ~(size_t)63), kSample, tmp, sizeof tmp, &m) == S2SR_OK && m >= kSample)
            return lzw_encode_literal(src, n, dst, cap, out_n);
    }
    return lzw_encode_dictionary(src, n, dst, cap, out_n);
}

static int lzw_encode_dictionary(const uint8_t* src, size_t n, uint8_t* dst, size_t cap, size_t* out_n) {
    // dictionary: open addressing over 32768 slots of (prefix code 12 | byte 8 | code 12) bits.  What costs is the probe
    // sequence of a MISS (on imagery most lookups miss: the string is new), so the table is kept nearly empty (load <= 0.12);
    // measured on a 786-KB strip of SR output: 8192 slots 7.7 ms, 16384 5.4, 32768 5.0 (r03's separate key / value arrays with
    // generation stamps, 16384 slots: 5.8).  A ClearCode wipes the table (128 KB per ~6.5 KB of incompressible input at worst).
    static const int HBITS = 15, HSIZE = 1 << HBITS;
    uint32_t tab[HSIZE];
    size_t pos = 0;
    uint64_t acc = 0;
    int nacc = 0;
    auto put = [&](int code, int width) -> bool {    // MSB-first; four bytes leave the accumulator at a time
        acc = (acc << width) | (uint32_t)code;
        nacc += width;
        if (nacc >= 32) {
            if (cap - pos < 4) {                     // the tail of a tight buffer: byte by byte
                while (nacc >= 8) {
                    if (pos >= cap) return false;
                    dst[pos++] = (uint8_t)(acc >> (nacc - 8));
                    nacc -= 8;
                }
                return true;
            }
            const uint32_t v = __builtin_bswap32((uint32_t)(acc >> (nacc - 32)));
            memcpy(dst + pos, &v, 4);
            pos += 4;
            nacc -= 32;
        }
        return true;
    };
    memset(tab, 0xFF, sizeof tab);                   // 0xFFFFFFFF: (prefix 4095, byte 255) -> code 4095 is never assigned
    auto reset = [&]() { memset(tab, 0xFF, sizeof tab); };
    int next = 258, width = 9;
    if (!put(256, width)) return S2SR_E_CAPACITY;
    if (n == 0) {
        if (!put(257, width)) return S2SR_E_CAPACITY;
    } else {
        int prefix = src[0];
        for (size_t i = 1; i < n; ++i) {
            const int k = src[i];
            const uint32_t key = ((uint32_t)prefix << 8) | (uint32_t)k;                 // 12 + 8 bits
            uint32_t h = (key * 2654435761u) >> (32 - HBITS);
            bool found = false;
            while (tab[h] != 0xFFFFFFFFu) {
                if ((tab[h] >> 12) == key) { found = true; break; }
                h = (h + 1) & (HSIZE - 1);
            }
            if (found) { prefix = (int)(tab[h] & 0xFFF); continue; }
            if (!put(prefix, width)) return S2SR_E_CAPACITY;
            tab[h] = (key << 12) | (uint32_t)next++;
            if (next == 4094) {                       // table full: ClearCode at the current width, start over
                if (!put(256, width)) return S2SR_E_CAPACITY;
                reset();
                next = 258; width = 9;
            } else if (next > (1 << width) - 1) {
                ++width;
            }
            prefix = k;
        }
        if (!put(prefix, width)) return S2SR_E_CAPACITY;
        ++next;                                       // the decoder adds an entry after this code too
        if (next == 4094) { if (!put(256, width)) return S2SR_E_CAPACITY; width = 9; }
        else if (next > (1 << width) - 1) ++width;
        if (!put(257, width)) return S2SR_E_CAPACITY;
    }
    while (nacc >= 8) {
        if (pos >= cap) return S2SR_E_CAPACITY;
        dst[pos++] = (uint8_t)(acc >> (nacc - 8));
        nacc -= 8;
    }
    if (nacc > 0) {
        if (pos >= cap) return S2SR_E_CAPACITY;
        dst[pos++] = (uint8_t)(acc << (8 - nacc));
    }
    *out_n = pos;
    return S2SR_OK;
}
