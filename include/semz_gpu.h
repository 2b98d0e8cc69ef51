/*
 * semz_gpu.h -- C ABI of the MI355X batched decode engine for se-mz/decompress.
 *
 * The reference is pure Common Lisp with no FFI (README.org:28); its "plugin
 * API" for a format is three generic functions (common.lisp:262-289):
 *
 *   byte-source->decompression-state (format byte-source &key)  -> state, header
 *   next-decompressed-chunk (state)                             -> chunk start end finalp
 *   make-reset-state (old-state)                                -> new-state | nil
 *
 * and all decode time is spent inside next-decompressed-chunk methods
 * (SURVEY.md section 3a).  This library replaces "one next-decompressed-chunk loop
 * per unit, serially" by "one device call for a batch of independent units".
 * The entry points below are what a CFFI binding in the Lisp host would bind
 * (INTEGRATION.md shows the stub); plain pointers and sizes only.
 *
 * A "unit" is one independently decodable piece:
 *   a gzip member      (gzip.lisp:31-115; independence: gzip.lisp:89-93)
 *   a zlib member      (zlib.lisp:21-78)
 *   a raw deflate member (deflate.lisp:365-433)
 *   a bzip2 block      (bzip2.lisp:351-395, 456-526)
 *   an xz block        (xz.lisp:260-457, lzma2.lisp:77-204)
 *   a raw LZMA2 stream (lzma2.lisp:206-218)
 *
 * Status codes: semz_status.h (one per `die` site on the path).
 * Return values of the functions: 0 on success, negative on API misuse or a
 * HIP failure (semz_last_error() has the text).  Data errors never make a call
 * fail: they are reported per unit in semz_result.status.
 */
#ifndef SEMZ_GPU_H
#define SEMZ_GPU_H

#include <stddef.h>
#include <stdint.h>
#include "semz_status.h"

#ifdef __cplusplus
extern "C" {
#endif

/* format keywords of the reference (example.lisp:77-78) */
enum {
    SEMZ_FMT_DEFLATE = 0, SEMZ_FMT_ZLIB = 1, SEMZ_FMT_GZIP = 2, SEMZ_FMT_BZIP2 = 3,
    SEMZ_FMT_RAW_LZMA = 4, SEMZ_FMT_LZMA = 5, SEMZ_FMT_RAW_LZMA2 = 6, SEMZ_FMT_LZMA2 = 7,
    SEMZ_FMT_XZ = 8
};

/* unit kinds */
enum {
    SEMZ_UNIT_DEFLATE = 0,     /* raw deflate member starting at in_off */
    SEMZ_UNIT_ZLIB = 1,        /* zlib member: 2(+4) byte header, deflate, Adler-32 BE */
    SEMZ_UNIT_GZIP = 2,        /* gzip member: header, deflate, CRC-32 + ISIZE */
    SEMZ_UNIT_BZIP2_BLOCK = 3, /* block whose 48-bit magic starts at (in_off, in_bit_off) */
    SEMZ_UNIT_XZ_BLOCK = 4,    /* LZMA2 payload of an xz block (+ filters, + check) */
    SEMZ_UNIT_LZMA2 = 5,       /* raw LZMA2 chunk sequence */
    SEMZ_UNIT_LZMA = 6,        /* raw LZMA data (:raw-lzma, lzma.lisp:809-835): range coder start at in_off;
                                  lc + lp <= 4 (what liblzma writes); wider literal contexts are
                                  SEMZ_E_UNSUPPORTED here and need the next kind */
    SEMZ_UNIT_LZMA_WIDE = 7    /* same for any lc <= 8, lp <= 4: the literal coder's probabilities live in
                                  HBM scratch (6 MiB per unit), several times slower per literal.  A batch that
                                  holds such a unit, or is submitted under this kind, runs all its units this way */
};

/* semz_unit.flags */
enum {
    SEMZ_UF_HAVE_DICT_FN = 1u << 0,  /* zlib: caller has a dictionary function (zlib.lisp:43-44) */
    SEMZ_UF_HAVE_IN_SIZE = 1u << 2,  /* xz: block header declares a compressed size */
    SEMZ_UF_HAVE_OUT_SIZE = 1u << 3  /* xz: block header declares an uncompressed size; lzma: size_out is valid */
};

/* xz filters that follow LZMA2 when decoding (semz_unit.filters, xz.lisp:311-344) */
enum { SEMZ_FILTER_END = 0, SEMZ_FILTER_DELTA = 1, SEMZ_FILTER_X86 = 2,
       /* the other branch/call/jump filters, xz-bcj-filters.lisp:45-167 */
       SEMZ_FILTER_POWERPC = 3, SEMZ_FILTER_IA64 = 4, SEMZ_FILTER_ARM = 5, SEMZ_FILTER_ARMTHUMB = 6,
       SEMZ_FILTER_SPARC = 7, SEMZ_FILTER_ARM64 = 8 };

/* xz check types (xz.lisp:12-31) */
enum { SEMZ_CHECK_NONE = 0, SEMZ_CHECK_CRC32 = 1, SEMZ_CHECK_CRC64 = 4, SEMZ_CHECK_SHA256 = 10 };

#define SEMZ_OUT_AUTO UINT64_MAX

typedef struct semz_unit {
    uint64_t in_off;      /* byte offset of the unit inside `in` */
    uint64_t in_limit;    /* one past the last byte this unit may read: reads beyond are
                             `eof` (io.lisp:93-104).  In all-members mode this is the end of
                             the input, because the reference's bit readers may look ahead
                             into the next member (interface.lisp:81-84). */
    uint64_t out_off;     /* byte offset of the unit's output inside `out`, or SEMZ_OUT_AUTO
                             (bzip2 blocks only): the engine packs the outputs of the batch
                             back to back in unit order and reports each offset in
                             semz_result.out_off; out_cap is then ignored per unit and the
                             call's out_cap bounds the whole batch */
    uint64_t out_cap;     /* bytes available at out_off; exceeding it is SEMZ_E_OUT_CAP */
    uint32_t kind;        /* SEMZ_UNIT_* */
    uint32_t flags;       /* SEMZ_UF_* */
    uint32_t in_bit_off;  /* bzip2: bit offset 0..7 (MSB-first) of the magic inside byte in_off */
    uint32_t param0;      /* deflate: window size (0 = 32768); bzip2: level 1..9; lzma/lzma2/xz: dictionary size */
    uint32_t param1;      /* xz: check type; lzma: lc | lp << 8 | pb << 16 */
    uint32_t param2;      /* xz: BCJ start offset (xz.lisp:326-331), shared by the BCJ stages of the chain;
                             lzma: eof-mode 0 :maybe, 1 :always, 2 :never (lzma.lisp:575-577, 824-825) */
    uint32_t param3;      /* xz: delta distances minus one, 8 bits per delta stage in chain order */
    uint32_t filters;     /* xz: filter chain applied to the LZMA2 output, 4 bits per stage
                             (SEMZ_FILTER_*), first applied stage in bits 3:0; in the block header
                             that is the filter listed just before LZMA2 (xz.lisp:311-312) */
    uint64_t prefix_off;  /* deflate/zlib preset history: span inside `in` (deflate.lisp:419-425) */
    uint64_t prefix_len;
    uint64_t size_in;     /* xz: declared compressed size (SEMZ_UF_HAVE_IN_SIZE) */
    uint64_t size_out;    /* xz: declared uncompressed size; lzma: :decompressed-size (SEMZ_UF_HAVE_OUT_SIZE) */
} semz_unit;

typedef struct semz_result {
    int32_t  status;      /* semz_status */
    uint32_t detail;      /* first argument of the reference's message (bad code, ...) */
    uint64_t in_end_bit;  /* bit offset inside `in` one past the last bit the unit consumed,
                             trailer included: unit k+1 starts at ceil(in_end_bit / 8) */
    uint64_t out_len;     /* bytes written at out_off */
    uint64_t out_off;     /* where the output was written (== unit.out_off unless SEMZ_OUT_AUTO) */
    uint8_t  check[32];   /* computed checksum as the little-endian byte string the reference
                             compares (crc32/adler32/bzip2 crc: 4, crc64: 8, sha256: 32) */
    uint32_t check_len;
    uint32_t aux;         /* zlib: dictionary id when FDICT; gzip: header length */
} semz_result;

typedef struct semz_ctx semz_ctx;

/* Opens the engine on one GPU (one context per process per GPU; calls on one
 * context must not overlap).  device = HIP device ordinal. */
int semz_open(int device, semz_ctx **ctx);
/* One input, several GPUs, inside the C call (the reference has no counterpart, its unit of
 * parallelism is the caller's thread): devices[0] is the primary device and behaves as a context from semz_open;
 * the stream calls (semz_decode_stream, semz_decode_stream_into, semz_stream_decode / read / release) split the
 * input once on the host, cut the predicted batch into contiguous ranges of about equal work (output bytes where
 * the sizes are known -- gzip ISIZE, the xz index --, compressed bytes otherwise), decode every range on its own
 * device, the primary included, from the bytes of that range only (one host thread and one HIP stream per device,
 * no data-path collective) and walk the merged results in the reference's order.  Units the walk has to decode
 * on demand, and a failed unit whose input span was cut at a range boundary, are decoded on the primary device,
 * which takes the whole input then and only then.  The batch calls (semz_decode_batch*) use the primary device only.  A device may be
 * named more than once (two contexts on one GPU).  semz_close closes all of them. */
int semz_open_multi(const int *devices, int ndev, semz_ctx **ctx);
int semz_device_count(const semz_ctx *ctx);        /* 1 for semz_open, ndev for semz_open_multi */
void semz_close(semz_ctx *ctx);
const char *semz_last_error(const semz_ctx *ctx);

/* Reference message text for a status (common.lisp:53-57 and the `die` sites). */
const char *semz_strerror(int status);

/* Batched decode, host buffers: uploads in[0..in_len) and the unit table,
 * runs the kernels, downloads out[0..out_cap) spans of the units and results.
 * Replaces the per-unit next-decompressed-chunk loops (deflate.lisp:365-410,
 * gzip.lisp:95-115, zlib.lisp:59-78, bzip2.lisp:456-526, xz.lisp:381-457). */
int semz_decode_batch(semz_ctx *ctx, const uint8_t *in, size_t in_len,
                      const semz_unit *units, size_t nunits,
                      uint8_t *out, size_t out_cap, semz_result *results);
/* All units of one call belong to one family: SEMZ_UNIT_DEFLATE/ZLIB/GZIP may
 * be mixed; SEMZ_UNIT_BZIP2_BLOCK alone; SEMZ_UNIT_XZ_BLOCK/LZMA2 may be mixed. */

/* Same, everything device-resident (d_* are device pointers of this ctx's GPU);
 * asynchronous on the context's stream until semz_sync().  `kind` names the
 * family of the batch (any SEMZ_UNIT_* value of that family).  This is the
 * call the benchmark times.
 * The unit table lives on the device, so the host cannot check it: every decode
 * kernel checks its unit's spans against in_len / out_cap first and reports
 * SEMZ_E_BAD_UNIT in that unit's result, without touching memory through the
 * unit, when in_off > in_limit, in_limit > in_len, out_off + out_cap > out_cap
 * (of the call), the prefix span leaves `in`, in_bit_off > 7 or the kind is not
 * of the batch's family.  d_in needs no padding: the kernels read whole aligned
 * 16-byte words around the spans, which never leave the pages the spans lie on. */
int semz_decode_batch_device(semz_ctx *ctx, uint32_t kind, const void *d_in, size_t in_len,
                             const semz_unit *d_units, size_t nunits,
                             void *d_out, size_t out_cap, semz_result *d_results);
/* No counterpart in the reference.  For callers that cannot allocate device
 * memory while they decode (code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc): semz_decode_batch_device for the deflate
 * family (SEMZ_UNIT_DEFLATE / ZLIB / GZIP) that calls no hipMalloc, hipFree or
 * other allocating routine.  The per-unit trailer table that connects the
 * inflate and checksum kernels lives in a static device array of the library
 * (64 bytes per unit, 4 MiB per GPU) instead of grow-only scratch; at most
 * SEMZ_NOALLOC_MAX_UNITS units per call, more is refused with -1 and a message.
 * All calls on one GPU share that array: calls that overlap on one GPU race.
 * Asynchronous on the context's stream like semz_decode_batch_device. */
#define SEMZ_NOALLOC_MAX_UNITS 65536
int semz_decode_batch_device_noalloc(semz_ctx *ctx, uint32_t kind, const void *d_in, size_t in_len,
                                     const semz_unit *d_units, size_t nunits,
                                     void *d_out, size_t out_cap, semz_result *d_results);
/* Options: "profile" = 1 records a HIP event pair around every kernel of later calls.
 * "window-size" = the :window-size keyword of the :raw-lzma2 format (lzma2.lisp:206-218; values
 * above 2^32-1 are clipped as there), used by semz_decode_stream(SEMZ_FMT_RAW_LZMA2 / _RAW_LZMA).
 * "lc", "lp", "pb", "decompressed-size" (< 0 = nil), "eof-mode" (0 :maybe, 1 :always, 2 :never) = the
 * keywords of the :raw-lzma format (lzma.lisp:809-825).  "lzma-honour-size" = 1 makes SEMZ_FMT_LZMA
 * honour the size its header declares (off = the reference as written).
 * "lzma-no-fast" = 1 decodes xz / LZMA2 units with the general kernel only (for measurements).
 * "bzip2-tables" = 1 / 2 decodes bzip2 blocks with the narrow / wide Huffman root tables whatever the batch
 * size (0, the default: wide up to 1024 blocks per call, narrow beyond; for tests and measurements).
 * "bzip2-overlap-blocks" = the least number of blocks in a part of a large bzip2 batch (default 2048; a batch of
 * at least two parts decodes one part beside the inverse BWT of the part before it; 0 = always one piece).
 * "batch-budget" = the device bytes (input, output, decoder scratch) a predicted batch of a stream call
 * may take; a larger one is decoded window by window into host memory -- straight into the caller's
 * buffer for semz_decode_stream_into -- with the upload of one window, the kernels of another and the
 * download of a third side by side (the reference's bounded chunks, common.lisp:255-257, and its rule that
 * nothing delivered is lost before an error, interface.lisp:91-133).  0 (the default) = three quarters of
 * the memory the device has free when the call starts. */
int semz_set_option(semz_ctx *ctx, const char *name, long value);
/* Reads "batch-budget", "profile", "window-size" back, and "windows": the number of windows the last stream
 * call on this context decoded (0 = its batch went in one piece), and "moved": how many pieces of such windows
 * the last read found in the caller's buffer at another place than the stream has them (the walk did not take
 * the prediction as it was) and moved, and "uploaded-bytes": the bytes of input the last stream call sent to
 * devices, all devices of a semz_open_multi context together (every device gets the bytes of its own range; the
 * whole input goes to the primary device only when the walk has to decode something by itself), and
 * "lzma-handed-back": how many xz / LZMA2 units the accept-only kernel has handed to the general kernel on this
 * context's device since the library was loaded (a well-formed block with ordinary properties never is; tests and
 * measurements take the difference around a call).  No counterpart in the reference. */
int semz_get_option(semz_ctx *ctx, const char *name, long *value);
/* Per-kernel device time of the calls since the last reset ("profile" on):
 * spans come in launch order; *name is a static string. */
int semz_profile_count(semz_ctx *ctx);
int semz_profile_get(semz_ctx *ctx, int i, const char **name, float *ms);
void semz_profile_reset(semz_ctx *ctx);
int semz_sync(semz_ctx *ctx);
/* The HIP stream the kernels run on (hipStream_t), for event timing. */
void *semz_stream(semz_ctx *ctx);

/* Device memory helpers so that hosts without a HIP binding (ctypes, CFFI) can
 * stage buffers. */
int semz_dev_alloc(semz_ctx *ctx, size_t bytes, void **dptr);
int semz_dev_free(semz_ctx *ctx, void *dptr);
int semz_dev_upload(semz_ctx *ctx, void *dptr, const void *src, size_t bytes);
int semz_dev_download(semz_ctx *ctx, void *dst, const void *dptr, size_t bytes);
/* HIP events on the context's stream: record two, read the elapsed ms. */
int semz_event_create(semz_ctx *ctx, void **ev);
int semz_event_record(semz_ctx *ctx, void *ev);
int semz_event_elapsed_ms(semz_ctx *ctx, void *ev_start, void *ev_stop, float *ms);
int semz_event_destroy(semz_ctx *ctx, void *ev);

/* ---- host splitter ----
 * Finds the independent units of a multi-unit input without decoding it.
 *   gzip/zlib/deflate: caller-supplied member offsets (offsets[noffsets], ascending,
 *     offsets[0] == 0), e.g. a BGZF index or a record table; with noffsets == 0 a
 *     gzip input is walked through its BGZF "BC" extra subfields (gzip.lisp:10-29);
 *     if it has none the whole input is one unit chain (see semz_decode_stream).
 *   bzip2: stream header (bzip2.lisp:425-449) + bit-granular scan for the 48-bit
 *     block / footer magics (bzip2.lisp:460-503); candidates are validated after
 *     decoding by chaining in_end_bit (semz_decode_stream does this).
 *   xz: stream header (xz.lisp:78-112), then footer -> index -> block offsets
 *     (xz.lisp:166-233), block headers parsed on the host (xz.lisp:260-379).
 * out_sizes (optional, noffsets entries) gives each member's output size when
 * the caller knows it (zlib/deflate records); gzip members take ISIZE.
 * The returned table is malloc'ed; release it with semz_free(). */
typedef struct semz_stream_info {
    int32_t  status;        /* container-level parse status (semz_status) */
    uint32_t detail;
    uint64_t total_out;     /* sum of out_cap over units when known, else 0 */
    uint32_t first_header[8];/* format-specific first-member header words, see INTEGRATION.md */
    uint32_t nmembers;      /* streams/members seen (bzip2, xz) */
    uint32_t reserved;      /* semz_split, deflate family: leading units whose output size is known */
} semz_stream_info;

int semz_split(int fmt, const uint8_t *in, size_t in_len,
               const uint64_t *offsets, const uint64_t *out_sizes, size_t noffsets,
               semz_unit **units, size_t *nunits, semz_stream_info *info);
void semz_free(void *p);

/* decompress-all for a fully buffered input (interface.lisp:188-208): split,
 * decode every unit on the GPU, verify the cross-unit invariants on the host
 * in unit order (bzip2 stream CRC fold bzip2.lisp:42-49,489-494; xz index and
 * footer xz.lisp:166-233; "unit k ends where unit k+1 starts"; trailing data
 * interface.lisp:125-130), first error wins.  *out is malloc'ed (semz_free).
 * res->status is the stream's status; res->out_len the total output (for a
 * failed stream: everything the reference delivers before it reaches the bad
 * data, interface.lisp:95-108 -- the complete members and the good units of the
 * member that failed); res->in_end_bit/8 the bytes consumed.  A failed stream's
 * result also carries the other arguments of the reference's message beside
 * res->detail (its first): res->check / check_len = the checksum the decoder
 * computed, where the message prints one ("wanted ~8,'0x, got ~8,'0x":
 * zlib.lisp:75, gzip.lisp:108, bzip2.lisp:493, 520, xz.lisp:209, 220, 404, 452);
 * res->aux = a second argument of another kind (deflate.lisp:330 the reach the
 * window allowed, deflate.lisp:384 the required NLEN, gzip.lisp:77 the real
 * header CRC-16, gzip.lisp:112 the real size, xz.lisp:342 the alignment).  The
 * stored check of an xz block (xz.lisp:452, first argument) is in the input
 * itself, the check_len bytes that end at in_end_bit/8.  all_members == 0
 * decodes the first member only (interface.lisp:149-186). */
int semz_decode_stream(semz_ctx *ctx, int fmt, const uint8_t *in, size_t in_len,
                       int all_members, uint8_t **out, size_t *out_len,
                       semz_result *res, semz_stream_info *info);

/* The same in steps, for hosts that own the output buffer (a Lisp vector, a
 * pinned buffer) or want the output member by member:
 *   semz_stream_decode   decodes like semz_decode_stream and keeps the decoded
 *                        stream in device memory behind a handle (release it
 *                        with semz_stream_release before semz_close of its
 *                        context; several handles may be alive at once)
 *   semz_stream_members  the members the walk completed, in order: where each
 *                        ends in the input and in the output.  This is what the
 *                        next-decompressed-chunk / make-reset-state methods of a
 *                        host binding hand out one member at a time
 *                        (interface.lisp:91-133: a member's chunks are delivered
 *                        before the next member's header is looked at).  The
 *                        array belongs to the handle.
 *   semz_stream_read     bytes [offset, offset + len) of the decoded stream into
 *                        caller memory, straight from the device
 *   semz_decode_stream_into = decode + read of everything + release, into a
 *                        caller-owned buffer: no malloc'ed intermediate, no second
 *                        copy.  Returns 1, with the size needed in *out_len and
 *                        nothing copied, when out_cap is too small. */
typedef struct semz_member {
    uint64_t in_end;      /* bytes of the input consumed when the member ended */
    uint64_t out_end;     /* bytes of the output produced when the member ended */
} semz_member;
typedef struct semz_decoded semz_decoded;
int semz_stream_decode(semz_ctx *ctx, int fmt, const uint8_t *in, size_t in_len, int all_members,
                       semz_decoded **decoded, semz_result *res, semz_stream_info *info);
int semz_stream_members(const semz_decoded *decoded, const semz_member **members, size_t *nmembers);
int semz_stream_read(semz_decoded *decoded, uint64_t offset, void *dst, size_t len);
void semz_stream_release(semz_decoded *decoded);
int semz_decode_stream_into(semz_ctx *ctx, int fmt, const uint8_t *in, size_t in_len, int all_members,
                            uint8_t *out, size_t out_cap, size_t *out_len,
                            semz_result *res, semz_stream_info *info);

#ifdef __cplusplus
}
#endif
#endif /* SEMZ_GPU_H */
