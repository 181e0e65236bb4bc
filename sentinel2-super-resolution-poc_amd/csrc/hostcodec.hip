// Host-only helpers behind the C ABI (no device code): the TIFF flavour of LZW, which the file
// glue around the path needs for the LZW GeoTIFFs the reference reads and writes
// (reference server/app/wow_sr.py:59-79 reads through rasterio; :138-151 writes compress="lzw").
#include <stdint.h>
#include <string.h>

#include "../../include/s2sr.h"

extern "C" int s2sr_tiff_lzw_decode(const uint8_t* src, size_t n, uint8_t* dst, size_t cap, size_t* out_n) {
    if (!src || !dst || !out_n) return S2SR_E_INVALID;
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
