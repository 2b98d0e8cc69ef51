// container.h -- host-side container parsing shared by split.cpp and stream.cpp.
//
// These are the parts of the reference that stay on the host (SURVEY.md 8a, row
// a27 and the header halves of a11/a13): they touch a few bytes per unit and
// decide where the independent units are.
//   gzip header walk / BGZF "BC" subfield     gzip.lisp:10-87
//   bzip2 stream header, 48-bit magic scan    bzip2.lisp:425-449, 460-503
//   xz stream header / block header / index / footer / padding
//                                             xz.lisp:78-233, 260-379
#pragma once
#include <cstdint>
#include <cstddef>
#include <vector>
#include "../../include/semz_gpu.h"

namespace semz {

uint32_t crc32_host(const uint8_t *p, size_t n);

// ---- gzip ----
struct GzipHeader {
    int status = SEMZ_OK;
    uint32_t detail = 0;
    uint64_t header_len = 0;          // bytes before the deflate data
    uint32_t textp = 0, mtime = 0, xfl = 0, os = 0;
    int64_t extra_off = -1, extra_len = 0, name_off = -1, name_len = 0, comment_off = -1, comment_len = 0;
    int64_t bgzf_size = -1;           // total member size from a "BC" subfield, -1 if absent
};
GzipHeader parse_gzip_header(const uint8_t *in, uint64_t pos, uint64_t limit);

// ---- bzip2 ----
constexpr uint64_t BZ_BLOCK_MAGIC = 0x314159265359ull;
constexpr uint64_t BZ_FOOTER_MAGIC = 0x177245385090ull;
struct BzCandidate {
    uint64_t bit;        // bit offset of the 48-bit magic
    bool footer;
    uint32_t level;      // level of the stream the scan believes it is in (blocks only)
};
// every occurrence of the two magics at any bit offset, in order
std::vector<BzCandidate> scan_bzip2(const uint8_t *in, uint64_t in_len);
// "BZh1".."BZh9" at byte pos: status OK + level, or the reference's error (bzip2.lisp:425-449)
int parse_bzip2_header(const uint8_t *in, uint64_t pos, uint64_t in_len, uint32_t *level, uint32_t *detail);
uint64_t read_bits_be(const uint8_t *in, uint64_t in_len, uint64_t bit, unsigned n);   // n <= 56, zero padded

// ---- xz ----
struct XzBlockHeader {
    int status = SEMZ_OK;
    uint32_t detail = 0;
    uint32_t detail2 = 0;             // second argument of the reference's message: the computed header CRC
                                      // (xz.lisp:404-405), the alignment the BCJ offset misses (xz.lisp:342)
    uint64_t header_size = 0;         // bytes including the size byte, excluding the CRC-32
    bool have_csize = false, have_usize = false;
    uint64_t csize = 0, usize = 0;
    uint32_t dict_size = 0;
    uint32_t filters = 0;             // semz_unit.filters
    uint32_t bcj_offset = 0;          // semz_unit.param2
    uint32_t delta_dists = 0;         // semz_unit.param3
};
// block header at `pos` (first byte = size byte, non-zero); checks in the reference's order
XzBlockHeader parse_xz_block_header(const uint8_t *in, uint64_t pos, uint64_t in_len);
// stream header at pos (skip_magic: the 6 magic bytes are already consumed, pos points after them)
int parse_xz_stream_header(const uint8_t *in, uint64_t pos, uint64_t in_len, bool skip_magic, uint32_t *check_type,
                           uint64_t *end, uint32_t *detail);
unsigned xz_check_size(uint32_t check_type);
// multibyte integer (xz.lisp:141-154); returns status, advances *pos
int xz_varint(const uint8_t *in, uint64_t *pos, uint64_t limit, uint64_t *value);

struct XzIndexedBlock {
    uint64_t header_pos;              // position of the block header's size byte
    uint64_t unpadded, usize;         // from the index
    uint32_t check_type;
};
// Walks the index of every stream from the end of the input backwards and lists the blocks in
// file order.  Returns false when anything looks inconsistent; the caller then decodes forward.
bool xz_blocks_from_index(const uint8_t *in, uint64_t in_len, std::vector<XzIndexedBlock> &blocks);

}  // namespace semz
