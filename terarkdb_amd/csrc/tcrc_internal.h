// Declarations shared by the translation units of libtcrc.so (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <chrono>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "../../include/tcrc.h"

namespace tcrc_internal {

// internal flag bits (above the public TCRC_F_* ones)
constexpr uint32_t kFlagSegments = 0x200u;  // the items are segments of longer buffers (LongArgs): results are shifted and XORed
constexpr uint32_t kFlagStampBefore = 0x400u;  // the result is also stored as LE32 six bytes before the buffer (log header)

// How a launch tells a waiting host thread that it is done without a hipStreamSynchronize (which costs
// about 3.5 us more than watching a word of pinned memory): the workgroups (or, for segments of long buffers,
// the buffers) count themselves off at `counter`, and whoever completes the count zeroes it again and writes
// `value` to `flag`.  Only for launches whose results go to memory that is not cached on the GPU (pinned host
// memory), or are written with system-scope stores: the host acts on them the moment it sees the flag.
struct DoneSignal {
  uint32_t* counter;  // device memory, zero before and after the launch
  uint32_t* flag;     // pinned host memory (its device address); NULL: no signal
  uint32_t value;
};

struct BatchParams {
  const uint8_t* base;
  const tcrc_desc* desc;  // NULL: uniform geometry below
  uint64_t n;
  uint64_t stride;
  uint32_t block_len;
  uint32_t uniform_init;
  const uint8_t* suffix;   // per-buffer suffix byte, or NULL
  int32_t uniform_suffix;  // >= 0: same suffix byte for all buffers
  uint32_t flags;
  uint32_t verify;         // 1: compare with expected
  uint32_t* out;
  const uint32_t* expected;
  uint8_t* ok;
  unsigned long long* counts;
  const unsigned long long* n_dev;  // when set, the batch size is read from device memory (<= n)
  int lanes;                 // lanes per buffer to use (4, 8, 16, 64), 0 = let the engine choose
  const uint32_t* tab_main;  // 1024 dwords: Z_D set
  const uint32_t* tab_aux;   // (1+log2 G) sets of 1024 dwords: Z_4, Z_16, Z_32, ...
  DoneSignal done;
};

constexpr int kGeoCount = 4;

constexpr uint32_t kLogBlockBytes = 32768;                     // db/log_format.h:37
constexpr uint32_t kLogHeader = 7, kLogRecyclableHeader = 11;  // db/log_format.h:40,44

// One header step of db/log_reader.cc:317-384, in two halves so that a walker can have the next header on its
// way while it deals with the current record.
// wal_fetch: bytes 4..10 of the header at pos (length, type, and the log number a recyclable header carries -
// fetched together with length and type when the bytes are there at all, not after the type has come back:
// the walk is a chain of dependent loads, and this halves the chain for recyclable logs).
struct WalHeader {
  uint32_t lo, hi;  // header bytes 4..7 and 8..11 as they lie in memory (byte 11 is payload already)
};
__device__ __forceinline__ WalHeader wal_fetch(const uint8_t* img, uint64_t pos, uint64_t end) {
  WalHeader h = {0u, 0u};
  const uint64_t remaining = end - pos;
  const uint8_t* p = img + pos;
  if (remaining >= 12u) {
    // one unaligned 8-byte load, and nothing done to the result here: whoever fetches ahead wants the load
    // in flight, not waited for
    uint2 w;
    __builtin_memcpy(&w, p + 4, 8);
    h.lo = w.x;
    h.hi = w.y;
  } else if (remaining >= kLogHeader) {  // the last bytes of a block: byte by byte, never past `end`
    h.lo = (uint32_t)p[4] | ((uint32_t)p[5] << 8) | ((uint32_t)p[6] << 16);
    if (remaining >= kLogRecyclableHeader) {
      h.lo |= (uint32_t)p[7] << 24;
      h.hi = (uint32_t)p[8] | ((uint32_t)p[9] << 8) | ((uint32_t)p[10] << 16);
    }
  }
  return h;
}
// wal_decode: returns false when the walk of the block ends (code says why); otherwise the record occupies
// [pos, pos + hs + len).
__device__ __forceinline__ bool wal_decode(const WalHeader& h, uint64_t pos, uint64_t end, bool last, uint64_t log_number,
                                           uint32_t* hs_out, uint32_t* len_out, uint32_t* code) {
  const uint64_t remaining = end - pos;
  if (remaining < kLogHeader) {
    // a block trailer (zero padding), or at the very end of the file a truncated header
    *code = remaining == 0 ? TCRC_WAL_END_CLEAN : (last ? TCRC_WAL_END_SHORT_AT_EOF : TCRC_WAL_END_TRAILER);
    return false;
  }
  const bool has11 = remaining >= kLogRecyclableHeader;
  const uint32_t type = (uint32_t)(int)(signed char)(h.lo >> 16);  // the reference reads it through a char
  const uint32_t len = h.lo & 0xffffu;
  const uint32_t lognum = (h.lo >> 24) | (h.hi << 8);
  uint32_t hs = kLogHeader;
  if (type >= 5u && type <= 8u) {
    hs = kLogRecyclableHeader;
    if (!has11) {
      *code = last ? TCRC_WAL_END_SHORT_AT_EOF : TCRC_WAL_END_TRAILER;
      return false;
    }
    if ((uint64_t)lognum != log_number) {
      *code = TCRC_WAL_END_OLD_RECORD;
      return false;
    }
  }
  if ((uint64_t)hs + len > remaining) {
    *code = last ? TCRC_WAL_END_BAD_LEN_AT_EOF : TCRC_WAL_END_BAD_LEN;
    return false;
  }
  if (type == 0u && len == 0u) {
    *code = TCRC_WAL_END_ZERO_RECORD;
    return false;
  }
  *hs_out = hs;
  *len_out = len;
  return true;
}
__device__ __forceinline__ bool wal_step(const uint8_t* img, uint64_t pos, uint64_t end, bool last,
                                         uint64_t log_number, uint32_t* hs_out, uint32_t* len_out,
                                         uint32_t* code) {
  return wal_decode(wal_fetch(img, pos, end), pos, end, last, log_number, hs_out, len_out, code);
}


// Long buffers: every buffer is cut from its END into segments of 2^seg_shift bytes (the first one may be
// shorter); item s of the launch is segment s, and the kernel works out which buffer it belongs to and
// where it lies from the buffer descriptors - no per-segment arrays exist anywhere.
constexpr int kLongInline = 64;         // up to this many buffers travel in the kernel arguments
constexpr int kLongLdsBuffers = 2048;   // buffers of one launch (their running segment counts sit in LDS)
constexpr int kPowMinShift = 8;         // smallest segment: 256 bytes
constexpr int kPowMaxShift = 30;        // largest: 1 GiB
constexpr int kPowEntries = 2048;       // table T_s holds x^(8 * 2^s * i) for i < 2048
constexpr int kPowTables = kPowMaxShift + 11 - kPowMinShift + 1;  // T_8 .. T_41: T_(s+11) is T_s's second level
struct alignas(16) LongArgs {
  tcrc_desc inl[kLongInline];  // first, and 16-byte aligned: the kernel fetches them from the argument block
  uint32_t inl_first[kLongInline];
  uint32_t nbuf;              // buffers of this launch (<= kLongLdsBuffers)
  uint32_t seg_shift;         // log2 of the segment length
  uint32_t first_base;        // value of seg_first[0] (running counts may continue from an earlier launch)
  const tcrc_desc* bufs;      // [nbuf] in device memory, or NULL: inl / inl_first above
  const uint32_t* seg_first;  // [nbuf + 1] segments before buffer b
  const uint32_t* pow_lo;     // T_seg_shift
  const uint32_t* pow_hi;     // T_(seg_shift + 11)
  uint32_t* acc;              // one 128-byte line per buffer: word 0 XOR of the shifted segment remainders,
                              // word 1 segments arrived.  All zero before and after a launch.
};
struct NoLongArgs {};

// One launch of the log scan (tcrc_engine.hip: k_wal_fused) over a slice of a log image.
constexpr uint32_t kWalPiece = 2048;          // spans longer than this are cut into pieces of this size (from their end)
constexpr uint32_t kWalMaxBlocks = 1u << 19;  // 32 KiB blocks per launch (the entry format keeps 19 bits of block index)
struct WalScanParams {
  const uint8_t* img;          // first byte of the slice (a 32 KiB block boundary of the file)
  uint64_t size;               // bytes in the slice (the last block may be short)
  uint64_t log_number;
  uint64_t block_base;         // index of the slice's first block in the whole file
  uint32_t nblocks;
  uint32_t tail;               // the slice ends where the file ends
  uint32_t chunk_shift;        // log2 of the blocks per chunk dealt to a workgroup at a time
  uint32_t tune;               // development aid, normally 0
  uint32_t sort_classes;       // k_wal_split: entries of a placer step go into the ring by length class
  tcrc_wal_block* blocks;      // [nblocks]
  unsigned long long* counts;  // the caller's TCRC_CNT_* block (accumulated into)
  unsigned long long* records_host;  // when not NULL: the launch's candidate records are added to this word of
                                     // pinned host memory by the last workgroup to finish
  unsigned long long* launch_words;  // with records_host: two device words, zero before and after a launch
                                     // (records so far, workgroups done)
  uint32_t pow_piece[64];      // [q * 4 + l]: x^(8 * kWalPiece * q + 8 l) mod P
  unsigned long long* trace;   // development aid, normally NULL: 32 timestamps (100 MHz) per workgroup, see k_wal_fused
  const uint32_t* tab_main;
  const uint32_t* tab_aux;
};

// One launch of log framing (tcrc_engine.hip: k_wal_frame).  The host's plan is a list of 16-byte entries, one per
// piece of at most kFramePiece payload bytes; a physical record (fragment) is one entry, or - above kFramePiece -
// up to eight neighbouring ones in ONE round of eight entries, cut from its end like the scan's pieces, so that
// the wave that takes the round combines them in registers.  Rounds are independent of each other.
//   word 0 (64 bits): source offset of the piece (40 bits) | piece length << 40 (13 bits) | pieces behind it in
//                     its fragment << 53 (3 bits) | first piece << 56 | last piece << 57 | valid << 58 |
//                     zero bytes in front of the fragment's header << 59 (4 bits: a block tail too short for a header)
//   word 1 (64 bits): image offset of the piece's first payload byte (40 bits) | fragment's payload length << 40
//                     (16 bits) | record type << 56 (4 bits)
constexpr uint32_t kFramePiece = 4096;
constexpr uint64_t kFrameOffMask = (1ull << 40) - 1;
constexpr uint64_t kFrameMaxOffset = 1ull << 40;
struct WalFrameEntry {
  uint64_t a, b;
};
struct WalFrameParams {
  const uint8_t* src;
  uint8_t* dst;
  const WalFrameEntry* entries;  // nrounds * 8
  uint64_t nrounds;
  uint32_t header_size;  // 7 or 11
  uint32_t log_number;   // low 32 bits, for the 11-byte header
  uint32_t pow_piece[64];  // as in WalScanParams
  uint8_t* sink;           // 4 KiB of writable scratch
  const uint32_t* tab_main;
  const uint32_t* tab_aux;
};

// a * b mod P (register convention bit 31 = x^0), 32 shift-and-xor steps.
__device__ __forceinline__ uint32_t gf2_mulmod_dev(uint32_t a, uint32_t b) {
  uint32_t prod = 0;
#pragma unroll 8
  for (int i = 31; i >= 0; --i) {
    prod ^= ((a >> i) & 1u) ? b : 0u;
    b = (b >> 1) ^ ((b & 1u) ? 0x82f63b78u : 0u);
  }
  return prod;
}

int fail(int code, const std::string& msg);

// Host-side construction of the lookup tables of one geometry (tcrc_engine.hip).
void build_tables(int G, std::vector<uint32_t>* main_tab, std::vector<uint32_t>* aux);

// Launches the batch CRC kernel for the geometry the context prefers.
int run_batch(tcrc_ctx* ctx, BatchParams p, void* stream);
// tcrc_dev_batch_stamp / tcrc_dev_batch_verify with a completion word (for results in pinned host memory)
int batch_stamp(tcrc_ctx* ctx, const void* d_base, const tcrc_desc* d_desc, uint64_t n, const uint8_t* d_suffix,
                uint32_t flags, uint32_t* d_out, void* stream, const DoneSignal& done);
int batch_verify(tcrc_ctx* ctx, const void* d_base, const tcrc_desc* d_desc, uint64_t n, const uint8_t* d_suffix,
                 uint32_t flags, const uint32_t* d_expected, uint32_t* d_out, uint8_t* d_ok, uint64_t* d_counts,
                 void* stream, const DoneSignal& done);
// Launches it over the p.n segments that la describes (p.base, p.out and p.n are used).
int run_segments(tcrc_ctx* ctx, BatchParams p, const LongArgs& la, void* stream);
// Launches the fused log scan (tables and powers are filled in here).
int run_wal_scan(tcrc_ctx* ctx, WalScanParams p, void* stream);
// Launches log framing over the plan's entries (tables, powers and sink are filled in here).  phase_dst: rows
// aligned to the image's lines instead of the source's.
int run_wal_frame(tcrc_ctx* ctx, WalFrameParams p, bool phase_dst, void* stream);
// The multipliers of the log pieces: [q * 4 + l] = x^(8 * kWalPiece * q + 8 l) mod P, worked out once per process.
const uint32_t* wal_piece_powers();
// Per-device set-up of the long-buffer path at tcrc_init (the context's GPU is current): the power tables.
int long_device_setup(tcrc_ctx* ctx);
// Stops the staging threads of the context and frees its pinned slots (tcrc_shutdown).
void stage_pool_destroy(tcrc_ctx* ctx);
// Gives the context's slice of the static accumulator lines back (tcrc_shutdown).
void long_device_release(tcrc_ctx* ctx);

// Makes the context's GPU the calling thread's current device for the rest of the scope and puts the
// caller's device back afterwards (a multi-GPU host must not find its current device changed by a call).
struct DeviceScope {
  int prev = -1;
  bool changed = false;
  hipError_t err = hipSuccess;
  explicit DeviceScope(int device) {
    err = hipGetDevice(&prev);
    if (err == hipSuccess && prev != device) {
      err = hipSetDevice(device);
      changed = err == hipSuccess;
    }
  }
  ~DeviceScope() {
    if (changed) (void)hipSetDevice(prev);
  }
};
#define TCRC_DEVICE_SCOPE(ctx_)                                                                        \
  tcrc_internal::DeviceScope device_scope_((ctx_)->device);                                            \
  if (device_scope_.err != hipSuccess)                                                                 \
  return tcrc_internal::fail(TCRC_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(device_scope_.err))

// Waits for the completion word of a launch (DoneSignal); a launch that takes longer than a few hundred
// microseconds, or that failed and will never write its word, ends in an ordinary stream synchronisation.
inline hipError_t wait_done(const uint32_t* h_flag, uint32_t value, hipStream_t s) {
  const auto give_up = std::chrono::steady_clock::now() + std::chrono::microseconds(300);
  for (;;) {
    for (int i = 0; i < 64; ++i) {
      if (__atomic_load_n(h_flag, __ATOMIC_ACQUIRE) == value) return hipSuccess;
      __builtin_ia32_pause();
    }
    if (std::chrono::steady_clock::now() >= give_up) return hipStreamSynchronize(s);
  }
}

// Synchronises the stream when the scope is left without dismiss(): copies issued from host memory that is
// about to be freed or reused (a caller's buffer, a local vector, the staging ring) must not stay in flight
// behind an error return.
struct StreamGuard {
  hipStream_t s;
  bool armed = true;
  explicit StreamGuard(hipStream_t st) : s(st) {}
  void dismiss() { armed = false; }
  ~StreamGuard() {
    if (armed) (void)hipStreamSynchronize(s);
  }
};

#define HIP_TRY(expr)                                                                                  \
  do {                                                                                                 \
    hipError_t e_ = (expr);                                                                            \
    if (e_ != hipSuccess)                                                                              \
      return tcrc_internal::fail(TCRC_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));     \
  } while (0)

}  // namespace tcrc_internal

// Grow-only device scratch owned by a context.
struct tcrc_scratch {
  void* p = nullptr;
  size_t bytes = 0;
};

// What one host-pointer call needs for itself: a stream, staging memory on the device, and a block of pinned,
// GPU-addressable host memory through which small calls pass descriptors, short inputs and results without
// copy commands ([data | descriptors | suffix bytes | outputs | expectations | verdicts]).
struct tcrc_lane {
  std::mutex mu;  // held for the duration of the call that leased the lane
  hipStream_t stream = nullptr;
  tcrc_scratch data, desc, out, aux, blocks, records;
  char* zc = nullptr;
  uint32_t* done_counter = nullptr;  // device word for the lane's launches that signal their completion
  uint32_t done_seq = 0;             // value the next such launch writes (never 0)
};

struct tcrc_stage_pool;
struct tcrc_ctx {
  int device = -1;
  int cus = 0;
  int forced_g = 0;
  int forced_wal_chunk_shift = -1;  // >= 0: set for experiments only (TCRC_WAL_CHUNK_SHIFT)
  uint32_t wal_tune = 0;  // development aid (TCRC_WAL_TUNE): bits that switch parts of k_wal_split off
  bool wal_sort = true;   // the placer lays a step's entries down by length class (TCRC_WAL_SORT=0: in list order)
  bool wal_split = true;  // log scan with walker and placer apart (TCRC_WAL_SPLIT=0: the one-wave walker)
  bool nontemporal = true;  // batch data is read once: keep it from displacing other lines in L2/MALL
  uint32_t* d_tab_main[tcrc_internal::kGeoCount] = {};
  uint32_t* d_tab_aux[tcrc_internal::kGeoCount] = {};
  std::string description;
  // log-scan scratch (tcrc_wal.hip): one scan at a time per context
  std::mutex wal_mu;
  tcrc_scratch wal_misc, wal_list_counts, wal_list_first, wal_list_temp, wal_frags;
  uint8_t* d_sink = nullptr;  // 4 KiB of device memory that lane groups without work store to (log framing)
  // log framing (tcrc_wal.hip): plans upload their piece lists on a stream of the context's own; the buffers
  // (pinned + device) of freed plans are kept for the next plan (guarded by frame_mu)
  std::mutex frame_mu;
  hipStream_t frame_stream = nullptr;
  struct FrameBuf {
    void* h = nullptr;
    void* d = nullptr;
    size_t bytes = 0;
  };
  std::vector<FrameBuf> frame_free;
  int frame_phase_dst = 0;  // TCRC_FRAME_PHASE: 0 = rows aligned to the source's lines (measured faster), 1 = to the image's
  void* wal_trace = nullptr;  // development aid: device buffer of 32 x 8 bytes per CU the next scans write timestamps to
  uint64_t wal_calls = 0;  // picks the record-counter slot of a scan (guarded by wal_mu)
  unsigned long long* wal_pin = nullptr;  // 32 words of pinned host memory the scan kernels report record counts into
  // host-pointer entry points (tcrc_host.hip): every call leases one of the context's lanes, so calls from
  // different threads (flush and compaction threads share one context) run side by side
  static constexpr int kMaxLanes = 16;
  std::mutex lanes_mu;               // guards the vector, not the lanes
  std::vector<tcrc_lane*> lanes;     // created on demand, freed at shutdown
  // long-buffer scratch (tcrc_long.hip): one call at a time per context.  Lock order: a lane's mutex, then
  // wal_mu, long_mu or stage_mu.
  std::mutex long_mu;
  tcrc_scratch long_in, long_acc;
  const uint32_t* d_pow = nullptr;  // kPowTables x kPowEntries words on this GPU (shared by its contexts, never freed)
  uint32_t* d_static = nullptr;     // the context's slice of the accumulator lines in this GPU's copy of the library's static scratch
  int static_slice = -1;            // which slice (long_device_setup)
  bool debug_noalloc_memset = false;  // TCRC_DEBUG_NOALLOC_MEMSET=1: clear the lines with hipMemsetAsync (development aid)
  uint32_t* long_flag = nullptr;    // pinned word the long-buffer kernel writes its completion to (long_mu)
  uint32_t long_seq = 0;
  bool poll_completion = true;      // TCRC_NO_POLL=1: always hipStreamSynchronize (for comparison)
  char* long_pin = nullptr;         // pinned staging for the descriptors of a call with more than kLongInline buffers
  // Layout constants of a lane's zero-copy block (tcrc_lane::zc)
  static constexpr size_t kZcData = (size_t)64 << 10;     // bounce area for short pageable inputs
  static constexpr size_t kZcMaxItems = 4096;
  // Staging for uploads from pageable memory (tcrc_host.hip): pinned slots and the threads that fill them, shared
  // by the lanes, made at the first large upload (stage_mu guards the making)
  std::mutex stage_mu;
  static constexpr int kStageSlots = 24;
  static constexpr size_t kStageSlotBytes = (size_t)4 << 20;
  struct tcrc_stage_pool* stage_pool = nullptr;
  // running totals (tcrc_get_stats)
  std::atomic<uint64_t> st_batch_launches{0}, st_batch_buffers{0}, st_host_calls{0}, st_host_bytes_in{0},
      st_host_nanos{0}, st_log_images{0}, st_log_bytes{0}, st_long_calls{0}, st_long_bytes{0};
  // Pinned host buffers handed back through tcrc_host_free, kept for the next tcrc_host_alloc:
  // pinning memory costs about as much as copying it, and callers allocate staging per call.
  std::mutex pin_mu;
  std::vector<std::pair<void*, size_t>> pin_free;  // (pointer, capacity)
  std::vector<std::pair<void*, size_t>> pin_live;  // handed out: capacity by pointer
  size_t pin_free_bytes = 0;
};

namespace tcrc_internal {
// Adds one host-pointer call and its wall time to the context's totals when it goes out of scope.
struct HostCallTimer {
  tcrc_ctx* ctx;
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  explicit HostCallTimer(tcrc_ctx* c) : ctx(c) {}
  ~HostCallTimer() {
    if (!ctx) return;
    ctx->st_host_calls.fetch_add(1, std::memory_order_relaxed);
    ctx->st_host_nanos.fetch_add(
        (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(),
        std::memory_order_relaxed);
  }
};

// Makes sure s holds at least `bytes` of device memory on the context's GPU.
int scratch_reserve(tcrc_ctx* ctx, tcrc_scratch* s, size_t bytes);
void scratch_release(tcrc_scratch* s);
}  // namespace tcrc_internal
