// Host-only helpers behind the C ABI (no device code): the TIFF flavour of LZW, which the file
// glue around the path needs for the LZW GeoTIFFs the reference reads and writes
// (reference server/app/wow_sr.py:59-79 reads through rasterio; :138-151 writes compress="lzw").
#include <stdint.h>
#include <string.h>

#include "../../include/s2sr.h"

extern "C" int s2sr_tiff_lzw_decode(const uint8_t* src, size_t n, uint8_t* dst, size_t cap, size_t* out_n) {
    if ((!src && n) || !dst || !out_n) return S2SR_E_INVALID;
    // TIFF 6.0 section 13: MSB-first codes of 9..12 bits, ClearCode 256, EndOfInformation 257, code
    // width grows one code early ("early change")
    static const int MAXC = 4096;
    uint16_t prefix[MAXC];
    uint8_t suffix[MAXC], first[MAXC];
    uint16_t length[MAXC];
    for (int i = 0; i < 256; ++i) { prefix[i] = 0; suffix[i] = first[i] = (uint8_t)i; length[i] = 1; }
    int next = 258, width = 9, old = -1;
    size_t pos = 0, bitpos = 0;
    const size_t nbits = n * 8;
    while (bitpos + width <= nbits) {
        const size_t byte = bitpos >> 3;
        uint32_t w = 0;
        for (int k = 0; k < 4; ++k) w = (w << 8) | (byte + k < n ? src[byte + k] : 0);
        const int code = (int)((w >> (32 - width - (bitpos & 7))) & ((1u << width) - 1));
        bitpos += width;
        if (code == 257) break;
        if (code == 256) { next = 258; width = 9; old = -1; continue; }
        if (old < 0) {                     // first code after a clear is a literal
            if (code > 255) return S2SR_E_INVALID;
            if (pos >= cap) break;
            dst[pos++] = (uint8_t)code;
            old = code;
            continue;
        }
        int entry = code;
        if (code >= next) {                // KwKwK: the string being defined right now
            if (code != next || next >= MAXC) return S2SR_E_INVALID;
            prefix[next] = (uint16_t)old; suffix[next] = first[old]; first[next] = first[old];
            length[next] = (uint16_t)(length[old] + 1);
            entry = next;
        }
        const size_t len = length[entry];
        const size_t room = cap - pos;
        // write the string back to front (clipped to the room left)
        {
            int c = entry;
            size_t i = len;
            while (i > 0) {
                --i;
                if (i < room) dst[pos + i] = suffix[c];
                c = prefix[c];
            }
        }
        pos += len < room ? len : room;
        if (code < next && next < MAXC) {  // add old + first(entry)
            prefix[next] = (uint16_t)old; suffix[next] = first[entry]; first[next] = first[old];
            length[next] = (uint16_t)(length[old] + 1);
        }
        if (next < MAXC) ++next;
        if (next + 1 >= (1 << width) && width < 12) ++width;
        old = code;
        if (pos >= cap) break;
    }
    *out_n = pos;
    return S2SR_OK;
}

// TIFF LZW encoder (the counterpart of the decoder above; reference writes compress="lzw" GeoTIFFs,
// server/app/wow_sr.py:138-151).  One strip per call, so strips compress in parallel on host threads.
// Worst case output is ~1.4x the input (12-bit codes for single bytes) + a few bytes: the caller
// sizes dst as n*3/2 + 16.
extern "C" int s2sr_tiff_lzw_encode(const uint8_t* src, size_t n, uint8_t* dst, size_t cap, size_t* out_n) {
    if ((!src && n) || !dst || !out_n) return S2SR_E_INVALID;
    static const int HBITS = 14, HSIZE = 1 << HBITS;   // 4x the 4096 codes; entries carry a generation stamp
    uint32_t hkey[HSIZE];                                // so a ClearCode does not cost a table wipe
    uint16_t hval[HSIZE];
    uint32_t gen = 1;
    size_t pos = 0;
    uint64_t acc = 0;
    int nacc = 0;
    auto put = [&](int code, int width) -> bool {
        acc = (acc << width) | (uint32_t)code;
        nacc += width;
        while (nacc >= 8) {
            if (pos >= cap) return false;
            dst[pos++] = (uint8_t)(acc >> (nacc - 8));
            nacc -= 8;
        }
        return true;
    };
    memset(hkey, 0, sizeof hkey);
    auto reset = [&]() {
        if (++gen == (1u << 12)) { memset(hkey, 0, sizeof hkey); gen = 1; }
    };
    int next = 258, width = 9;
    if (!put(256, width)) return S2SR_E_CAPACITY;
    if (n == 0) {
        if (!put(257, width)) return S2SR_E_CAPACITY;
    } else {
        int prefix = src[0];
        for (size_t i = 1; i < n; ++i) {
            const int k = src[i];
            const uint32_t key = (gen << 20) | ((uint32_t)prefix << 8) | (uint32_t)k;   // 12 + 12 + 8 bits
            uint32_t h = ((((uint32_t)prefix << 8) | (uint32_t)k) * 2654435761u) >> (32 - HBITS);
            bool found = false;
            while ((hkey[h] >> 20) == gen) {
                if (hkey[h] == key) { found = true; break; }
                h = (h + 1) & (HSIZE - 1);
            }
            if (found) { prefix = hval[h]; continue; }
            if (!put(prefix, width)) return S2SR_E_CAPACITY;
            hkey[h] = key; hval[h] = (uint16_t)next++;
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
    if (nacc > 0) {
        if (pos >= cap) return S2SR_E_CAPACITY;
        dst[pos++] = (uint8_t)(acc << (8 - nacc));
    }
    *out_n = pos;
    return S2SR_OK;
}
