/*
 * semz_status.h -- per-unit status codes of the batched decode engine.
 *
 * One code per `die` call site of the reference on the hot path (SURVEY.md
 * section 8a/8b): the Lisp glue (or the Python host mirror) turns a code back
 * into the reference's `decompression-error` message via semz_strerror().
 * SEMZ_EOF maps to the `eof` condition (common.lisp:43-57).
 *
 * This header is shared by the product library (decompress_amd/csrc) and by
 * the test oracle (oracle/); it contains no code.
 */
#ifndef SEMZ_STATUS_H
#define SEMZ_STATUS_H

typedef enum semz_status {
    SEMZ_OK = 0,
    SEMZ_EOF = 1,                      /* common.lisp:56-57 "Premature end of input." */

    /* --- Huffman (huffman.lisp:182-184) --- */
    SEMZ_E_HUFF_UNDERFULL = 10,        /* "Underfull Huffman tree." */
    SEMZ_E_HUFF_OVERFULL = 11,         /* "Overfull Huffman tree." */

    /* --- Deflate (deflate.lisp) --- */
    SEMZ_E_DEFLATE_BTYPE = 20,         /* :397 "Block uses reserved BTYPE." */
    SEMZ_E_DEFLATE_STORED_LEN = 21,    /* :383-385 "Checksum mismatch in uncompressed block" */
    SEMZ_E_DEFLATE_CODELEN_BOUNDS = 22,/* :186-187 "Code lengths expand beyond bounds." */
    SEMZ_E_DEFLATE_REPEAT_FIRST = 23,  /* :189-190 "Tried to repeat non-existent last code length." */
    SEMZ_E_DEFLATE_DIST_CODE = 24,     /* :297-298 "Distance code out of bounds (0-29)" */
    SEMZ_E_DEFLATE_SINGLE_DIST = 25,   /* :299-305 "Unique distance code is out of bounds..." */
    SEMZ_E_DEFLATE_LEN_IN_LITONLY = 26,/* :323-324 "Length code in literal-only block" */
    SEMZ_E_DEFLATE_DISTANCE = 27,      /* :329-331 "Reference points back further ... than the window allows" */
    SEMZ_E_DEFLATE_LITLEN_CODE = 28,   /* :334-336 "Invalid literal/length code" */

    /* --- zlib (zlib.lisp) --- */
    SEMZ_E_ZLIB_METHOD = 40,           /* :30-31 */
    SEMZ_E_ZLIB_CINFO = 41,            /* :32-33 */
    SEMZ_E_ZLIB_FCHECK = 42,           /* :34-35 "Invalid flag checksum." */
    SEMZ_E_ZLIB_NO_DICT_FN = 43,       /* :43-44 */
    SEMZ_E_ZLIB_UNKNOWN_DICT = 44,     /* :47-48 unrecognized-zlib-dictionary */
    SEMZ_E_ZLIB_CHECKSUM = 45,         /* :74-76 "Invalid data checksum" */

    /* --- gzip (gzip.lisp) --- */
    SEMZ_E_GZIP_MAGIC = 50,            /* :60-62 */
    SEMZ_E_GZIP_METHOD = 51,           /* :63-64 */
    SEMZ_E_GZIP_RESERVED_FLAGS = 52,   /* :65-66 */
    SEMZ_E_GZIP_EXTRA_FIELDS = 53,     /* :24-25 "Inconsistent lengths for gzip extra fields." */
    SEMZ_E_GZIP_HEADER_CRC = 54,       /* :76-78 */
    SEMZ_E_GZIP_DATA_CRC = 55,         /* :107-108 "Invalid data checksum" */
    SEMZ_E_GZIP_ISIZE = 56,            /* :111-113 "Invalid data length check" */

    /* --- bzip2 (bzip2.lisp) --- */
    SEMZ_E_BZ_MAGIC = 60,              /* :429-430 */
    SEMZ_E_BZ_VERSION = 61,            /* :434-435 */
    SEMZ_E_BZ_LEVEL = 62,              /* :436-437 */
    SEMZ_E_BZ_BLOCK_MAGIC = 63,        /* :502-503 "Unrecognized magic bytes on block boundary." */
    SEMZ_E_BZ_NO_SYMBOLS = 64,         /* :358-359 */
    SEMZ_E_BZ_TREE_COUNT = 65,         /* :380-381 */
    SEMZ_E_BZ_NO_SELECTORS = 66,       /* :383-384 */
    SEMZ_E_BZ_SELECTOR = 67,           /* :306-307 "Invalid selector" */
    SEMZ_E_BZ_CODELEN = 68,            /* :336-337 "Code length goes out of bounds." */
    SEMZ_E_BZ_SELECTORS_EXHAUSTED = 69,/* :249-250 */
    SEMZ_E_BZ_AB_RUN_LENGTH = 70,      /* :261-262 "A/B run cannot possibly stay in bounds." */
    SEMZ_E_BZ_AB_RUN_SIZE = 71,        /* :271-272 "A/B run expands beyond block size." */
    SEMZ_E_BZ_TOO_MANY_CODES = 72,     /* :282-283 */
    SEMZ_E_BZ_BLOCK_CRC = 73,          /* :519-521 */
    SEMZ_E_BZ_STREAM_CRC = 74,         /* :492-494 */
    SEMZ_E_BZ_ORIGPTR = 75,            /* deliberate divergence: reference does not validate
                                          origPtr < n (bzip2.lisp:158); see DESIGN.md */

    /* --- LZMA (lzma.lisp) --- */
    SEMZ_E_LZMA_FIRST_BYTE = 80,       /* :57-58 "LZMA data doesn't begin with a zero octet." */
    SEMZ_E_LZMA_CODE_START = 81,       /* :60-61 "Code starts out of range." */
    SEMZ_E_LZMA_CODE_RANGE = 82,       /* :72-73, :187-188 "Code goes out of range." */
    SEMZ_E_LZMA_DICT_SIZE = 83,        /* :604-605 "Match extends beyond dictionary size." */
    SEMZ_E_LZMA_NO_DATA = 84,          /* :606-607 "Match extends beyond available data." */
    SEMZ_E_LZMA_MATCH_SIZE = 85,       /* :608-609 "Match extends beyond declared decompressed size." */
    SEMZ_E_LZMA_LITERAL_SIZE = 86,     /* :717-718 "Decompressed data goes beyond declared size." */
    SEMZ_E_LZMA_NO_EOF_MARKER = 87,    /* :642-643 */
    SEMZ_E_LZMA_ILLEGAL_EOF = 88,      /* :646, :730 "Illegal EOF marker in LZMA data." */
    SEMZ_E_LZMA_CONTINUES = 89,        /* :647 "LZMA data continues beyond declared size." */
    SEMZ_E_LZMA_NO_END = 90,           /* :659 */
    SEMZ_E_LZMA_FINISH_STATE = 91,     /* :773-774 "Range decoder finishes in invalid state." */
    SEMZ_E_LZMA_EARLY_END = 92,        /* :894 "Decompressed data ends before declared size." */
    SEMZ_E_LZMA_PB = 93,               /* :841-842 */

    /* --- LZMA2 (lzma2.lisp) --- */
    SEMZ_E_LZMA2_LCLP = 100,           /* :33-37 */
    SEMZ_E_LZMA2_DICT_BYTE = 101,      /* :42 */
    SEMZ_E_LZMA2_NEED_DICT_RESET = 102,/* :98, :160 */
    SEMZ_E_LZMA2_NEED_PROPS = 103,     /* :100 */
    SEMZ_E_LZMA2_CHUNK_OVERRUN = 104,  /* :116 "Embedded LZMA data goes beyond LZMA2 block size." */
    SEMZ_E_LZMA2_CONTROL = 105,        /* :149 "Illegal LZMA2 control byte" */
    SEMZ_E_LZMA2_CHUNK_SHORT = 106,    /* :201 "Embedded LZMA data is shorter than declared." */
    SEMZ_E_LZMA2_NO_STATE = 107,       /* reference crashes with a non-decompression error
                                          (lzma2.lisp:107 on a nil state); see DESIGN.md */

    /* --- XZ (xz.lisp) --- */
    SEMZ_E_XZ_MAGIC = 120,             /* :90-91 */
    SEMZ_E_XZ_RESERVED_BYTE = 121,     /* :92-93 */
    SEMZ_E_XZ_HEADER_CRC = 122,        /* :99-100 */
    SEMZ_E_XZ_RESERVED_FLAGS = 123,    /* :101-102 */
    SEMZ_E_XZ_CHECK_TYPE = 124,        /* :103-104 */
    SEMZ_E_XZ_TRAILING_GARBAGE = 125,  /* :133 */
    SEMZ_E_XZ_PADDING = 126,           /* :134 */
    SEMZ_E_XZ_VARINT_ZERO = 127,       /* :150-151 "Overlong multibyte integer encoding." */
    SEMZ_E_XZ_VARINT_LONG = 128,       /* :154 "Overlong multibyte integer." */
    SEMZ_E_XZ_INDEX_PADDING = 129,     /* :189-190 */
    SEMZ_E_XZ_INDEX_COUNT = 130,       /* :193-196 */
    SEMZ_E_XZ_INDEX_SIZES = 131,       /* :198-203 */
    SEMZ_E_XZ_INDEX_CRC = 132,         /* :208-210 */
    SEMZ_E_XZ_FOOTER_CRC = 133,        /* :219-221 */
    SEMZ_E_XZ_BACKWARD_SIZE = 134,     /* :222-225 */
    SEMZ_E_XZ_FOOTER_FLAGS = 135,      /* :226-228 */
    SEMZ_E_XZ_FOOTER_MAGIC = 136,      /* :229-230 */
    SEMZ_E_XZ_BLOCK_RESERVED = 137,    /* :266-267 */
    SEMZ_E_XZ_BLOCK_TRUNCATED = 138,   /* :287-288 */
    SEMZ_E_XZ_BLOCK_HDR_PADDING = 139, /* :290-291 */
    SEMZ_E_XZ_BLOCK_OVERRUN = 140,     /* :308-309 "Embedded compressed data goes beyond declared XZ block size." */
    SEMZ_E_XZ_DELTA_LAST = 141,        /* :315-316 */
    SEMZ_E_XZ_DELTA_PROPS = 142,       /* :317-318 */
    SEMZ_E_XZ_BCJ_LAST = 143,          /* :324-325 */
    SEMZ_E_XZ_BCJ_PROPS = 144,         /* :329-331 */
    SEMZ_E_XZ_BCJ_ALIGN = 145,         /* :341-342 */
    SEMZ_E_XZ_LZMA2_NOT_LAST = 146,    /* :347-348 */
    SEMZ_E_XZ_LZMA2_PROPS = 147,       /* :349-350 */
    SEMZ_E_XZ_LZMA2_RESERVED = 148,    /* :351-352 */
    SEMZ_E_XZ_FILTER = 149,            /* :377 "Unrecognized filter type" */
    SEMZ_E_XZ_BLOCK_HDR_CRC = 150,     /* :403-405 */
    SEMZ_E_XZ_TOTAL_SIZE = 151,        /* :419-421 */
    SEMZ_E_XZ_BLOCK_SHORT = 152,       /* :428-430 "XZ block is smaller than its declared size." */
    SEMZ_E_XZ_BLOCK_OUT_SIZE = 153,    /* :432-435 */
    SEMZ_E_XZ_BLOCK_PADDING = 154,     /* :441-442 */
    SEMZ_E_XZ_BLOCK_CHECK = 155,       /* :451-454 "Incorrect block checksum" */
    SEMZ_E_XZ_STREAM_TOO_LONG = 156,   /* :87 */

    /* --- interface.lisp --- */
    SEMZ_E_TRAILING_DATA = 170,        /* interface.lisp:130 "Trailing data detected." */

    /* --- engine-level conditions with no counterpart in the reference --- */
    SEMZ_E_OUT_CAP = 200,              /* unit produced more than its out_cap: retry with a larger span */
    SEMZ_E_UNSUPPORTED = 201,          /* unit uses a feature outside the current hot-path scope */
    SEMZ_E_NOT_RUN = 202,              /* unit was not decoded (an earlier unit in its chain failed) */
    SEMZ_E_NO_MEMORY = 204,            /* semz_decode_stream: the size the data asks for (a declared size, or what a
                                          unit keeps expanding to) does not fit the device memory the engine can get;
                                          the reference streams in bounded memory and would go on (common.lisp:255-257) */
    SEMZ_E_BAD_UNIT = 203              /* semz_decode_batch_device: the unit's spans leave the buffers of the call
                                          (in_limit > in_len, out_off + out_cap > out_cap, prefix outside `in`),
                                          or its kind does not belong to the batch's family; nothing was read or written */
} semz_status;

#endif /* SEMZ_STATUS_H */
