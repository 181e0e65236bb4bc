// Pieces of the PNG / deflate encoder shared by the host encoder (pngenc.hip) and the device tile encoder (pngdev.hip).
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace s2sr {
namespace png {

// Token alphabet of both encoders: 0..255 a literal byte, 256 + (len - 3) a match of `len` bytes (3..258) at distance 1.
struct BlockCode {
    uint32_t tb[512];           // token -> code bits (a match: length code | extra bits | the 1-bit distance code 0) | bit count << 24
    uint32_t eob;               // end-of-block code, same packing
    uint8_t header[640];        // BFINAL, BTYPE = 10, HLIT / HDIST / HCLEN and the run-length coded code lengths, LSB first
    uint32_t header_bits;
    uint64_t body_bits;         // what the counted tokens + the end-of-block code take
};
// Huffman code of one block from its token histogram (512 slots, end-of-block implied once).
void build_block_code(const uint32_t* tok_freq, bool final_block, BlockCode* bc);

uint32_t crc32_update(uint32_t crc, const uint8_t* p, size_t n);
uint32_t adler32_update(uint32_t adler, const uint8_t* p, size_t n);
bool write_file(const char* path, const uint8_t* data, size_t n);     // makes missing parent directories
struct Piece { const uint8_t* p; size_t n; };
bool write_file_pieces(const char* path, const Piece* pieces, int count);   // the same from up to 8 pieces (one writev)

}  // namespace png
}  // namespace s2sr
