/*
 * tcrc - batched masked-CRC32C engine for AMD Instinct MI355X (gfx950).
 *
 * C ABI of libtcrc.so.  Plain pointers and sizes only.  The library replaces the
 * checksum arithmetic on TerarkDB's per-record path; each entry point names the
 * reference interface it stands in for (paths relative to the reference tree).
 *
 * Conventions
 *   - "d_" arguments are device pointers valid on the context's GPU; "h_" are host.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Device
 *     entry points only enqueue work: results are ready once the stream is synced.
 *   - return value: TCRC_OK or a negative TCRC_ERR_* code; tcrc_last_error() gives
 *     text.  A checksum MISMATCH is data, never an error code.
 *   - CRC values follow the reference: Extend(init, data, n) takes and returns
 *     finalised CRCs (util/crc32c.h:23-29); masked values are Mask() of those
 *     (util/crc32c.h:38-41).
 */
#ifndef TCRC_H_
#define TCRC_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TCRC_OK 0
#define TCRC_ERR_INVALID (-1)   /* bad argument */
#define TCRC_ERR_NO_DEVICE (-2) /* no usable GPU / HIP runtime failure at init */
#define TCRC_ERR_HIP (-3)       /* a HIP call failed */
#define TCRC_ERR_TOO_LARGE (-4) /* size beyond what the entry point supports */

typedef struct tcrc_ctx tcrc_ctx;

/* One buffer of a batch: bytes [offset, offset+len) of the batch's base pointer,
 * checksummed as Extend(init, ...).  init = 0 gives Value().  len may be up to 1 GiB
 * (longer inputs: tcrc_dev_extend_long, tcrc_extend).  The host-pointer entry points
 * reject a longer descriptor with TCRC_ERR_TOO_LARGE; the device-pointer ones cannot
 * see the descriptors, so there the kernel reads nothing for such a descriptor, its
 * output value is unspecified and a verify counts it as a mismatch.  Offsets in device-resident
 * descriptors are trusted like any pointer handed to a kernel: a span outside the caller's allocation is
 * an out-of-bounds access on the GPU. */
typedef struct tcrc_desc {
  uint64_t offset;
  uint32_t len;
  uint32_t init;
} tcrc_desc;

/* flags */
#define TCRC_F_MASK 0x1u          /* outputs are Mask(crc) instead of crc */
#define TCRC_F_EXPECT_AFTER 0x2u  /* verify: stored value is the LE32 right after the buffer (SST trailer), or right
                                     after the suffix byte when one is given - where TCRC_F_STORE_AFTER puts it */
#define TCRC_F_EXPECT_BEFORE 0x4u /* verify: stored value is the LE32 six bytes before the buffer (WAL header) */
#define TCRC_F_STORE_AFTER 0x8u   /* stamp: also write the block trailer into the image, right after the buffer: the
                                     suffix byte (if one is given) and then the result as LE32 - the 5-byte
                                     [type][checksum] of table/block_based_table_builder.cc:626-631.  The image must
                                     be writable there and the trailers must not overlap other buffers. */

#define TCRC_F_SHORT_SPANS 0x10u  /* descriptor lists: most buffers are shorter than about 1.4 KiB.  Results do not
                                     depend on it; the engine then puts four lanes on a buffer instead of eight
                                     (sixteen buffers to a wave), which is faster for such buffers and slower for
                                     longer ones.  Uniform blocks need no hint: their length is known. */

/* verify result block (device memory, 3 x uint64): */
#define TCRC_CNT_MISMATCH 0  /* number of buffers whose checksum did not match */
#define TCRC_CNT_FIRST_BAD 1 /* smallest index of a mismatching buffer, ~0 if none */
#define TCRC_CNT_CHECKED 2   /* number of buffers checked */
#define TCRC_CNT_WORDS 3

/* ---- lifecycle ------------------------------------------------------------- */
/* Creates an engine on GPU `device` (uploads the lookup tables).  Replaces the
 * reference's static-init choice of implementation (util/crc32c.cc:1203-1225). */
int tcrc_init(int device, tcrc_ctx** out);
int tcrc_shutdown(tcrc_ctx* ctx);
const char* tcrc_last_error(void);
/* Text like the reference's IsFastCrc32Supported() (util/crc32c.cc:445-481). */
const char* tcrc_describe(tcrc_ctx* ctx);

/* Running totals of one context: the engine's counterpart of the reference's block_checksum_time perf
 * counter (table/block_fetcher.cc:40) and WAL/ SST tickers (monitoring/statistics.cc).  All values only grow
 * until tcrc_reset_stats. */
typedef struct tcrc_stats {
  uint64_t batch_launches; /* launches of the batch CRC kernel */
  uint64_t batch_buffers;  /* buffers handed to it (the capacity, for lists counted on the device) */
  uint64_t host_calls;     /* calls of the host-pointer entry points (tcrc_host_*, tcrc_extend, tcrc_value) */
  uint64_t host_bytes_in;  /* bytes they copied to the device */
  uint64_t host_nanos;     /* wall time spent inside them */
  uint64_t log_images;     /* log images scanned (tcrc_dev_wal_verify) */
  uint64_t log_bytes;      /* and their bytes */
  uint64_t long_calls;     /* tcrc_dev_extend_long calls */
  uint64_t long_bytes;     /* and the bytes they covered */
  uint64_t host_lanes;     /* lanes (stream + staging) the context has created for concurrent host calls: not reset */
} tcrc_stats;
int tcrc_get_stats(tcrc_ctx* ctx, tcrc_stats* out);
int tcrc_reset_stats(tcrc_ctx* ctx);

/* ---- scalar helpers (host arithmetic on 32-bit values only) ------------------ */
/* util/crc32c.h:38-47 */
uint32_t tcrc_mask(uint32_t crc);
uint32_t tcrc_unmask(uint32_t masked);

/* ---- device-resident batches ------------------------------------------------ */
/* n_blocks buffers of block_len bytes at d_base + i*stride, all with the same init
 * and (if suffix_byte >= 0) one extra trailing byte covered by the CRC; d_suffix,
 * when not NULL, gives that byte per block instead.  This is the SST trailer rule
 * crc = Extend(Value(block), &type, 1); Mask(crc)
 * (table/block_based_table_builder.cc:598-601) over a whole batch. */
int tcrc_dev_blocks_stamp(tcrc_ctx* ctx, const void* d_base, uint64_t n_blocks, uint32_t block_len, uint64_t stride,
                          uint32_t init, int suffix_byte, const uint8_t* d_suffix, uint32_t flags, uint32_t* d_out,
                          void* stream);

/* Same geometry, but compares against stored masked values: d_expected[i], or the
 * in-image LE32 when TCRC_F_EXPECT_AFTER/_BEFORE is set.  Writes d_ok[i] = 1/0 when
 * d_ok != NULL, the computed masked values when d_out != NULL, and accumulates
 * d_counts (TCRC_CNT_*; zero/~0-initialise with tcrc_dev_counts_reset).  Batch form
 * of BlockFetcher::CheckBlockChecksum (table/block_fetcher.cc:36-71). */
int tcrc_dev_blocks_verify(tcrc_ctx* ctx, const void* d_base, uint64_t n_blocks, uint32_t block_len, uint64_t stride,
                           uint32_t init, int suffix_byte, const uint8_t* d_suffix, uint32_t flags,
                           const uint32_t* d_expected, uint32_t* d_out, uint8_t* d_ok, uint64_t* d_counts,
                           void* stream);

/* Arbitrary buffers described by d_desc[0..n).  Batch form of
 * crc32c::Extend(init, data, n) (util/crc32c.cc:1226-1228), with the optional
 * per-buffer suffix byte and Mask as above. */
int tcrc_dev_batch_stamp(tcrc_ctx* ctx, const void* d_base, const tcrc_desc* d_desc, uint64_t n,
                         const uint8_t* d_suffix, uint32_t flags, uint32_t* d_out, void* stream);
int tcrc_dev_batch_verify(tcrc_ctx* ctx, const void* d_base, const tcrc_desc* d_desc, uint64_t n,
                          const uint8_t* d_suffix, uint32_t flags, const uint32_t* d_expected, uint32_t* d_out,
                          uint8_t* d_ok, uint64_t* d_counts, void* stream);

/* ---- long buffers ------------------------------------------------------------------ */
/* d_out[i] = Extend(h_desc[i].init, d_base + h_desc[i].offset, h_desc[i].len) for buffers of
 * hundreds of KiB up to 1 GiB each: every buffer is cut into segments of segment_len bytes
 * (0 = choose; otherwise a power of two from 256 to 2^30) that are checksummed in parallel and merged
 * with a GF(2) shift-and-xor - one kernel launch per call (per 2048 buffers); up to 64 descriptors travel
 * in the kernel arguments, more are copied to the GPU first.  Data is on the GPU, the (few) descriptors are on the host.
 * Values are plain CRCs (apply tcrc_mask on the host if a masked value is wanted).
 * Stands in for crc32c::Extend on large inputs (util/crc32c.cc:1226-1228; callers such as
 * utilities/backupable/backupable_db.cc:1434-1449).  Returns with d_out complete: with up to 64 buffers
 * the call watches a completion word the kernel writes to pinned memory once every result has been written
 * through to device memory (the stream may still show the kernel as running for a moment; whatever is
 * enqueued on it afterwards is ordered behind the kernel as usual); with more it synchronises the stream. */
int tcrc_dev_extend_long(tcrc_ctx* ctx, const void* d_base, const tcrc_desc* h_desc, uint64_t n, uint32_t segment_len,
                         uint32_t* d_out, void* stream);

/* tcrc_dev_extend_long without the wait, for callers with many such calls in a row (whole-file checksums buffer
 * after buffer, utilities/backupable/backupable_db.cc:1434-1449): the kernel is enqueued on `stream` and the
 * call returns; d_out is complete once the stream has run, and only the last call of a row needs waiting for.
 * At most 64 buffers per call (TCRC_ERR_TOO_LARGE beyond).  The partial results live in lines the context owns
 * (allocated at the first long-buffer call): the enqueued calls of one context must run one after the other -
 * one stream, or events between streams - and must not be mixed with tcrc_dev_extend_long calls still in flight. */
int tcrc_dev_extend_long_enqueue(tcrc_ctx* ctx, const void* d_base, const tcrc_desc* h_desc, uint64_t n,
                                 uint32_t segment_len, uint32_t* d_out, void* stream);

/* tcrc_dev_extend_long for callers that cannot allocate (code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  Same arguments and results,
 * with these limits: at most 64 buffers per call (TCRC_ERR_TOO_LARGE beyond), each up to 1 GiB.  The
 * call only enqueues two kernels (one clears the lines) on `stream`: it allocates nothing, issues no copy
 * command and does not synchronise, so d_out is complete once the stream has run (the caller synchronises).
 * The partial results live in the library's static device memory (each GPU holds its own copy for the life
 * of the process: 64 KiB of accumulator lines in eight slices of 8 KiB, and the 272 KiB of power tables built
 * at tcrc_init).  Every context takes one slice at tcrc_init - the least used one, so up to eight contexts
 * on a GPU have lines of their own and their calls may overlap freely.  Calls of ONE context (or of
 * contexts that share a slice, from the ninth context of a GPU on) share its 8 KiB: if they overlap -
 * different streams without an event between them, different threads - they race and give wrong values;
 * keep a context's calls on one stream, or order them with events. */
int tcrc_dev_extend_long_noalloc(tcrc_ctx* ctx, const void* d_base, const tcrc_desc* h_desc, uint64_t n,
                                 uint32_t segment_len, uint32_t* d_out, void* stream);

/* ---- write-ahead-log images ------------------------------------------------------ */
/* How the structural walk of one 32 KiB log block ended (db/log_reader.cc:314-384). */
#define TCRC_WAL_END_CLEAN 0          /* records fill the block exactly */
#define TCRC_WAL_END_TRAILER 1        /* fewer than a header's worth of bytes left: block trailer */
#define TCRC_WAL_END_SHORT_AT_EOF 2   /* same, but the file ends here: truncated header (kBadHeader) */
#define TCRC_WAL_END_BAD_LEN 3        /* header_size + length runs past the block (kBadRecordLen) */
#define TCRC_WAL_END_BAD_LEN_AT_EOF 4 /* same in the file's last block (kBadHeader / kEof) */
#define TCRC_WAL_END_ZERO_RECORD 5    /* type 0 and length 0: preallocated space (kBadRecord) */
#define TCRC_WAL_END_OLD_RECORD 6     /* recyclable header with another log number (kOldRecord) */

/* Result for one 32 KiB block of a log image. */
typedef struct tcrc_wal_block {
  uint32_t n_records;  /* records that pass every structural check of ReadPhysicalRecord */
  uint32_t first_bad;  /* smallest index of a record whose CRC does not match, 0xffffffff if none */
  uint32_t end_offset; /* offset inside the block where the walk stopped */
  uint32_t end_code;   /* TCRC_WAL_END_* */
} tcrc_wal_block;

/* Verifies every physical record of a complete log image resident on the GPU the way
 * log::Reader::ReadPhysicalRecord does with checksum=true (db/log_reader.cc:314-406):
 * d_blocks[b] describes block b = bytes [32768*b, 32768*(b+1)) of the image.  Within a
 * block the reference stops at the first bad checksum, so first_bad is what its policy
 * layer needs.  d_counts accumulates TCRC_CNT_* over all candidate records (TCRC_CNT_FIRST_BAD
 * is the index of the first 32 KiB block that holds a mismatching record).  One kernel
 * launch per 16 GiB of image.  With h_records NULL the call only enqueues on `stream`;
 * otherwise it waits for the stream and *h_records receives the number of candidate
 * records (the kernel hands it over through pinned host memory, no copy command). */
int tcrc_dev_wal_verify(tcrc_ctx* ctx, const void* d_img, uint64_t size, uint64_t log_number, tcrc_wal_block* d_blocks,
                        uint64_t* d_counts, uint64_t* h_records, void* stream);

/* The same for one part of a log that is verified piece by piece (a file streamed through a staging
 * buffer): d_part holds the bytes from block `first_block` of the file on, `size` of them - whole 32 KiB
 * blocks when more_follows != 0, in which case the part's last block is not treated as the end of the
 * file (db/log_reader.cc:276-312 only reports a truncated header or record at the real end).
 * d_blocks[b] describes block first_block + b; TCRC_CNT_FIRST_BAD counts blocks from the file's start. */
int tcrc_dev_wal_verify_part(tcrc_ctx* ctx, const void* d_part, uint64_t size, uint64_t log_number,
                             uint64_t first_block, int more_follows, tcrc_wal_block* d_blocks, uint64_t* d_counts,
                             uint64_t* h_records, void* stream);

/* One physical record of a log image, as ReadPhysicalRecord would meet it. */
#define TCRC_WAL_REC_BAD 0     /* checksum mismatch: the reference drops the rest of the block here */
#define TCRC_WAL_REC_OK 1      /* checksum matches and no earlier record of the block failed */
#define TCRC_WAL_REC_SKIPPED 2 /* lies after a bad record of the same block: never examined by the reference */
typedef struct tcrc_wal_record {
  uint64_t offset;      /* of the record header in the image */
  uint32_t length;      /* payload bytes (the header's length field) */
  uint8_t type;         /* record type byte (db/log_format.h:19-35) */
  uint8_t status;       /* TCRC_WAL_REC_* */
  uint16_t header_size; /* 7, or 11 for the recyclable types */
} tcrc_wal_record;

/* Lists the physical records of a log image in file order: per 32 KiB block the
 * (offset, length, type, verdict) sequence that ReadPhysicalRecord (db/log_reader.cc:314-406) walks,
 * cut where the block's walk ends.  d_blocks is the result of tcrc_dev_wal_verify for the same image
 * and log number.  Up to `capacity` records are written to d_records; *h_count receives the total
 * (call with capacity 0 to size the array).  Returns with the list complete. */
int tcrc_dev_wal_list(tcrc_ctx* ctx, const void* d_img, uint64_t size, uint64_t log_number,
                      const tcrc_wal_block* d_blocks, tcrc_wal_record* d_records, uint64_t capacity,
                      uint64_t* h_count, void* stream);

/* Frames logical records that are already in device memory into a log image in device memory, the
 * way n calls of log::Writer::AddRecord append them (db/log_writer.cc:45-151): fragments that never
 * cross a 32 KiB block, 7-byte headers (11 with the log number when recycle != 0), zero-filled block
 * tails shorter than a header, and Mask(crc32c(type | [log number] | payload)) in every header.
 * ONE kernel: every payload byte is read once, stored to its place in the image and folded into its
 * fragment's checksum from the same registers; where the headers go is planned on the host from the
 * record lengths alone.
 * Record i is the h_recs[i].len bytes at d_src + h_recs[i].offset (h_recs[i].init is ignored).
 * block_offset is the position inside the current 32 KiB block at which the image continues an
 * existing log (0 for a new file); *block_offset_out (may be NULL) is that position after the call
 * (always below 32768: 0 when the image ends exactly at a block boundary), ready to be passed to the next call.
 * *h_written receives the image size; tcrc_wal_framed_size gives it beforehand.  d_dst must not
 * overlap d_src.  Returns with the image complete (plan, enqueue, wait for the stream). */
int tcrc_dev_wal_frame(tcrc_ctx* ctx, const void* d_src, const tcrc_desc* h_recs, uint64_t n, uint64_t log_number,
                       int recycle, uint32_t block_offset, void* d_dst, uint64_t dst_capacity, uint64_t* h_written,
                       uint32_t* block_offset_out, uint64_t* h_physical_records, void* stream);
/* Bytes tcrc_dev_wal_frame will write for these records (host arithmetic on the lengths only). */
uint64_t tcrc_wal_framed_size(const tcrc_desc* h_recs, uint64_t n, int recycle, uint32_t block_offset);

/* The same in two steps, for a writer that keeps the GPU busy: the plan is host work (the loop of
 * db/log_writer.cc:45-100 over the record lengths; its piece list goes to the device on a stream of the
 * context's own), the enqueue is one kernel launch on the caller's stream and waits for nothing - so a group
 * commit can plan batch k+1 while batch k is being framed.  A plan fixes the record lengths and source offsets,
 * the header kind, the log number and the starting block offset; it can be enqueued any number of times (other
 * source bytes at the same offsets, other destinations).  The image is complete once the stream has run.
 * tcrc_wal_plan_free waits for the launches that read the plan.  Limits: source offsets and the image below 1 TiB. */
typedef struct tcrc_wal_plan tcrc_wal_plan;
int tcrc_wal_plan_create(tcrc_ctx* ctx, const tcrc_desc* h_recs, uint64_t n, uint64_t log_number, int recycle,
                         uint32_t block_offset, tcrc_wal_plan** out);
uint64_t tcrc_wal_plan_image_bytes(const tcrc_wal_plan* plan);       /* what tcrc_wal_framed_size gives */
uint32_t tcrc_wal_plan_block_offset(const tcrc_wal_plan* plan);      /* block offset after the image */
uint64_t tcrc_wal_plan_physical_records(const tcrc_wal_plan* plan);  /* headers the image holds */
int tcrc_dev_wal_frame_enqueue(tcrc_ctx* ctx, tcrc_wal_plan* plan, const void* d_src, void* d_dst,
                               uint64_t dst_capacity, void* stream);
/* By default every enqueue records an event behind its launch, so that tcrc_wal_plan_free can wait for the plan's
 * last reader whenever it is called; between launches queued back to back that event costs about 3.5 us (113 against
 * 110 us per launch for 256 MiB of 2040-byte records).  A caller that frames in a loop and waits for its stream
 * anyway can take that over: after tcrc_wal_plan_caller_waits(plan, 1) no event is recorded, and the plan must not
 * be freed before the work enqueued with it from then on has finished. */
int tcrc_wal_plan_caller_waits(tcrc_wal_plan* plan, int yes);
void tcrc_wal_plan_free(tcrc_wal_plan* plan);

/* Sets d_counts to {0, ~0, 0}. */
int tcrc_dev_counts_reset(tcrc_ctx* ctx, uint64_t* d_counts, void* stream);

/* ---- host-memory batches --------------------------------------------------------- */
/* The reference's path starts and ends in host memory (log file buffers, blocks read through
 * env/io_posix.cc).  These entry points take host pointers, copy the bytes to the GPU, run the
 * device path above and copy the results back; they return when the results are in place.
 * Calls from different threads on one context run side by side: each takes one of the context's lanes (a
 * stream, staging memory on the device and a pinned block for small calls), of which a context creates up to
 * 16 on demand - the reference calls this path from several flush and compaction threads at once
 * (table/block_based_table_builder.cc:580, db/db_impl_write.cc:844-854).  Further callers wait for a lane. */
void* tcrc_host_alloc(tcrc_ctx* ctx, size_t bytes); /* pinned memory: copies from it run at link rate */
void tcrc_host_free(tcrc_ctx* ctx, void* p);
int tcrc_host_batch_stamp(tcrc_ctx* ctx, const void* h_base, uint64_t base_bytes, const tcrc_desc* h_desc, uint64_t n,
                          const uint8_t* h_suffix, uint32_t flags, uint32_t* h_out);
int tcrc_host_blocks_stamp(tcrc_ctx* ctx, const void* h_base, uint64_t n_blocks, uint32_t block_len, uint64_t stride,
                           uint32_t init, int suffix_byte, uint32_t flags, uint32_t* h_out);
/* Host form of tcrc_dev_blocks_verify (BlockFetcher::CheckBlockChecksum, table/block_fetcher.cc:36-71, over an
 * extent of equally spaced blocks as read from a file): stored values from h_expected[i], or with
 * TCRC_F_EXPECT_AFTER from the image, right behind every block (behind its suffix byte when one is given - the
 * image then extends 4 or 5 bytes beyond the last block).  h_out / h_ok may be NULL. */
int tcrc_host_blocks_verify(tcrc_ctx* ctx, const void* h_base, uint64_t n_blocks, uint32_t block_len, uint64_t stride,
                            uint32_t init, int suffix_byte, uint32_t flags, const uint32_t* h_expected, uint32_t* h_out,
                            uint8_t* h_ok, uint64_t* h_counts);
/* h_counts: 3 x uint64 (TCRC_CNT_*), filled by the call. */
int tcrc_host_batch_verify(tcrc_ctx* ctx, const void* h_base, uint64_t base_bytes, const tcrc_desc* h_desc, uint64_t n,
                           const uint8_t* h_suffix, uint32_t flags, const uint32_t* h_expected, uint32_t* h_out,
                           uint8_t* h_ok, uint64_t* h_counts);
/* Host form of tcrc_dev_wal_verify: h_blocks gets ceil(size/32768) entries. */
int tcrc_host_wal_verify(tcrc_ctx* ctx, const void* h_img, uint64_t size, uint64_t log_number, tcrc_wal_block* h_blocks,
                         uint64_t* h_counts, uint64_t* h_records);

/* Host form of tcrc_dev_wal_verify_part.  h_counts is ACCUMULATED into (initialise it to {0, ~0, 0} before
 * the first part), so that the parts of one file add up to what tcrc_host_wal_verify reports for it. */
int tcrc_host_wal_verify_part(tcrc_ctx* ctx, const void* h_part, uint64_t size, uint64_t log_number,
                              uint64_t first_block, int more_follows, tcrc_wal_block* h_blocks, uint64_t* h_counts,
                              uint64_t* h_records);

/* Host form of verify + list: h_blocks (may be NULL) as in tcrc_host_wal_verify, then the record list. */
int tcrc_host_wal_list(tcrc_ctx* ctx, const void* h_img, uint64_t size, uint64_t log_number, tcrc_wal_block* h_blocks,
                       tcrc_wal_record* h_records, uint64_t capacity, uint64_t* h_count);

/* The reference's scalar interface on host bytes: *out = crc32c::Extend(init, data, n)
 * (util/crc32c.cc:1226-1228) and crc32c::Value(data, n) (util/crc32c.h:29), computed on the GPU
 * (copy in, long-buffer kernels, 4 bytes back).  Unlike the reference's functions these can fail
 * (no GPU, HIP error), so the value comes back through *out.  Any n; inputs above 256 MiB are fed
 * in pieces chained through init.  One call costs a PCIe round trip: this is for whole files and
 * large buffers (utilities/backupable/backupable_db.cc:1260-1263,1434-1449), not for one record. */
int tcrc_extend(tcrc_ctx* ctx, uint32_t init, const void* h_data, size_t n, uint32_t* out);
int tcrc_value(tcrc_ctx* ctx, const void* h_data, size_t n, uint32_t* out);

/* ---- the reference's own scalar signature ------------------------------------------ */
/* uint32_t (*)(uint32_t, const char*, size_t): the type of the pointer the reference dispatches
 * crc32c::Extend through (`typedef uint32_t (*Function)(...)`, util/crc32c.cc:439; `ChosenExtend`,
 * util/crc32c.cc:1225-1228), so an integration can write `ChosenExtend = tcrc_crc32c_extend;`.
 * Returns crc32c::Extend(init, data, n), computed on the GPU of a process-wide default context that the
 * first call creates (device from the environment variable TCRC_DEVICE, default 0); calls are serialised
 * inside that context, short inputs take the path without copy commands.  One call is a device round trip
 * (about 16 us), so this is for completeness of the boundary, not for per-record use.
 *
 * The signature leaves no room for an error, and the library holds no CPU implementation of the checksum.
 * tcrc_set_scalar_fallback hands in the function to call instead when there is no GPU or the engine fails -
 * the reference's own ExtendImpl<...> / crc32c_3way, i.e. what ChosenExtend pointed to before.  Without a
 * fallback such a call prints the engine's error and aborts the process.  tcrc_scalar_fallback_calls counts
 * the calls that went to the fallback. */
typedef uint32_t (*tcrc_extend_fn)(uint32_t init, const char* data, size_t n);
uint32_t tcrc_crc32c_extend(uint32_t init, const char* data, size_t n);
void tcrc_set_scalar_fallback(tcrc_extend_fn fn);
uint64_t tcrc_scalar_fallback_calls(void);

/* ---- several GPUs ----------------------------------------------------------------------- */
/* The path shards by units that carry their own checksums (SST blocks, 32 KiB log blocks, files), so a
 * verification corpus splits into contiguous ranges that are verified independently; what comes together at the
 * end is three counters.  The loop this parallelises is DBImpl::VerifyChecksum (db/db_impl.cc:4657-4690: one
 * table file after the other) over BlockBasedTable::VerifyChecksumInBlocks
 * (table/block_based_table_reader.cc:2647-2695: one block after the other); the reference has no counterpart
 * of the entry points themselves (it is one host thread).
 *
 * Contiguous range [*lo, *hi) of unit `index` of `count` shards of n_units (sizes differ by at most one). */
void tcrc_shard_range(uint64_t n_units, uint32_t index, uint32_t count, uint64_t* lo, uint64_t* hi);

/* One process, several GPUs: a group owns one context and one stream per member; devices[i] is member i's GPU
 * (a GPU may be named more than once).  A group call runs every member's shard on a host thread of its own and
 * reduces {mismatches, first bad, checked} on the host - no collective. */
typedef struct tcrc_group tcrc_group;
int tcrc_group_init(const int* devices, int n, tcrc_group** out);
int tcrc_group_shutdown(tcrc_group* g);
int tcrc_group_size(const tcrc_group* g);
tcrc_ctx* tcrc_group_ctx(tcrc_group* g, int member); /* for the single-context entry points, e.g. to fill a shard */
int tcrc_group_device(const tcrc_group* g, int member);
/* Uniform blocks resident on the members' GPUs: member m holds the blocks tcrc_shard_range(n_blocks, m, members) of
 * the corpus at d_bases[m] (and their stored values at d_expected[m], unless TCRC_F_EXPECT_AFTER).  h_counts
 * (3 x uint64) receives the corpus's totals; TCRC_CNT_FIRST_BAD is a block index of the corpus.  Returns when
 * every member is done. */
int tcrc_group_dev_blocks_verify(tcrc_group* g, const void* const* d_bases, uint64_t n_blocks, uint32_t block_len,
                                 uint64_t stride, uint32_t init, int suffix_byte, uint32_t flags,
                                 const uint32_t* const* d_expected, uint64_t* h_counts);
/* The same for one image in host memory: every member uploads and verifies its range (the host link of every
 * GPU works at once). */
int tcrc_group_host_blocks_verify(tcrc_group* g, const void* h_base, uint64_t n_blocks, uint32_t block_len,
                                  uint64_t stride, uint32_t init, int suffix_byte, uint32_t flags,
                                  const uint32_t* h_expected, uint8_t* h_ok, uint64_t* h_counts);

/* One process per GPU: the ranks' only exchange.  d_counts is the rank's TCRC_CNT_* block on its GPU, counted in
 * the rank's own numbering; first_unit_of_rank is the corpus index of the rank's first unit.  Enqueues on
 * `stream`: ncclAllReduce(ncclSum) over {mismatches, checked} and ncclAllReduce(ncclMin) over the first bad unit's
 * corpus index, in place, on the caller's communicator (an ncclComm_t passed as void*) - every rank's block then
 * holds the corpus's totals.  librccl is opened at the first of these calls (the copy already loaded into the
 * process, if there is one), so libtcrc.so itself does not depend on it.
 * The three helpers let a caller without RCCL headers make the communicator: rank 0 gets the 128-byte id and
 * hands it to the others by whatever means the processes share (a file, MPI, a key-value store). */
int tcrc_rccl_get_unique_id(void* id128);
int tcrc_rccl_comm_init_rank(tcrc_ctx* ctx, int nranks, const void* id128, int rank, void** comm_out);
int tcrc_rccl_comm_destroy(void* comm);
int tcrc_rccl_allreduce_counts(tcrc_ctx* ctx, void* comm, uint64_t* d_counts, uint64_t first_unit_of_rank, void* stream);

/* ---- measurement aid ----------------------------------------------------------- */
/* Streams nbytes from d_base with loads only (no checksum work) so a benchmark can report
 * the read rate this device actually reaches beside the CRC kernels' rate.  d_scratch
 * must hold 2 * 1024 * (number of CUs) uint32.  No counterpart in the reference. */
int tcrc_dev_probe_read(tcrc_ctx* ctx, const void* d_base, uint64_t nbytes, uint32_t* d_scratch, uint32_t wgs_per_cu,
                        void* stream);

/* The lookup tables the kernels use for `lanes` lanes per buffer (4, 8, 16 or 64), as built on the host
 * at tcrc_init: main_out gets the 1024 words of Z_D (D = 16 * lanes), aux_out the reduction sets and the
 * inverse-shift multipliers; *aux_words their count.  Needs no GPU.  For tests that model the kernel's
 * arithmetic on the CPU; the reference's counterpart is its static table0_..table3_ (util/crc32c.cc:50-313). */
int tcrc_debug_tables(int lanes, uint32_t* main_out, uint32_t* aux_out, size_t aux_capacity, size_t* aux_words);

#ifdef __cplusplus
}
#endif
#endif /* TCRC_H_ */
