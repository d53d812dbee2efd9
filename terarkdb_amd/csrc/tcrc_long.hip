// Long buffers (hundreds of KiB to GiBs): the CRC of one buffer is computed by many lane
// groups at once.  A buffer is cut, from its END, into segments of S bytes (the first one
// may be shorter); every segment is walked by a lane group of the batch kernel as a buffer
// of its own, and the segment remainders are merged with the GF(2) identity
//     R(A || B) = R(A) * x^(8|B|)  xor  R(B)        (mod P)
// i.e. a segment followed by q whole segments is multiplied by x^(8 S q) and XORed into the
// buffer's result.  This is the role the reference's three-stream combine plays inside one
// call of crc32c_3way (util/crc32c.cc:641-658,686-699), widened from 3 streams to thousands;
// the caller-visible function is still crc32c::Extend (util/crc32c.cc:1226-1228), e.g. for
// the backup engine's whole-file checksums (utilities/backupable/backupable_db.cc:1434-1449).
//
// One call is ONE kernel launch: the kernel derives every segment from the buffer
// descriptors (which travel in the kernel arguments when there are at most kLongInline of
// them), the buffers' accumulator lines clean themselves (the lane group that delivers a
// buffer's last segment takes the value out, zeroes the line and writes the result), and
// the powers x^(8 S q) come from tables built once per GPU.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <string>
#include <vector>

#include "gf2.h"
#include "tcrc_internal.h"

using namespace tcrc_internal;

namespace {

// ---- static device memory -------------------------------------------------------------------
// [accumulator lines of the entry point that allocates nothing | power tables T_8 .. T_41]
// Every GPU the library runs on holds its own copy (at its own address) until the process ends.
// The accumulator lines come in kStaticSlices slices of kLongInline lines: every context of a GPU takes a
// slice of its own at tcrc_init (the ninth and later contexts share with earlier ones), so that calls of
// different contexts do not meet in the same lines.
constexpr uint32_t kStaticSlices = 8;
constexpr uint32_t kSliceWords = (uint32_t)kLongInline * 32u;  // one 128-byte line per buffer
constexpr uint32_t kStaticAccWords = kStaticSlices * kSliceWords;
constexpr uint32_t kStaticPowWords = (uint32_t)kPowTables * (uint32_t)kPowEntries;
constexpr uint32_t kStaticWords = kStaticAccWords + kStaticPowWords;
static __device__ __align__(256) uint32_t scratch[kStaticWords];

constexpr int kMaxDevices = 64;
std::mutex g_static_mu;
uint32_t* g_static_ptr[kMaxDevices] = {};  // device address of `scratch` on every GPU set up so far
uint32_t g_slice_users[kMaxDevices][kStaticSlices] = {};  // contexts holding each slice (guarded by g_static_mu)

// T_s[i] = x^(8 * 2^s * i) mod P for s = kPowMinShift + blockIdx.x, i < kPowEntries.
__global__ __launch_bounds__(1024) void k_pow_tables(uint32_t* tables) {
  const uint32_t s = (uint32_t)kPowMinShift + blockIdx.x;
  uint32_t base = 0x40000000u;  // x; squared s + 3 times: x^(2^(s+3)) = x^(8 * 2^s)
  for (uint32_t k = 0; k < s + 3u; ++k) base = gf2_mulmod_dev(base, base);
  for (uint32_t i = threadIdx.x; i < (uint32_t)kPowEntries; i += 1024u) {
    uint32_t r = 0x80000000u, sq = base;
    for (uint32_t e = i; e; e >>= 1) {
      if (e & 1u) r = gf2_mulmod_dev(r, sq);
      sq = gf2_mulmod_dev(sq, sq);
    }
    tables[(size_t)blockIdx.x * kPowEntries + i] = r;
  }
}

// Zeroes `words` dwords.  The library clears device words that a caller's stream may find under graph capture with
// this kernel and never with hipMemsetAsync: recorded into a HIP graph, the memset node fills zeros on the graph's
// first launch only - from the second launch on it writes a 16-byte pattern of stale words (ROCm 7.2; the fill
// runs in its place in front of the next kernel, its value is what is wrong), so the accumulator lines held
// garbage, no buffer's arrival count reached its total and no result was written.
__global__ __launch_bounds__(256) void k_zero_words(uint32_t* p, uint32_t words) {
  for (uint32_t i = threadIdx.x; i < words; i += 256u) p[i] = 0u;
}

// Segment length for a call.  512 bytes up to 2 MiB in all: short segments spread a small call over many waves
// (what the waves of a workgroup deliver to a buffer is folded in LDS before it reaches the buffer's accumulator
// line, so the arrivals that queue there number one per workgroup).  Then 1 KiB and 2 KiB, which keep a call of
// up to 8 MiB within 512 rounds of eight segments - 64 workgroups of eight waves: a segment costs some 700
// instructions beside its rows, so more and shorter segments than the chip needs to be busy only add work
// (4 MiB in one buffer: 19.0 us with 512-byte segments, 16.5 us with 1 KiB).  Beyond that the shortest power of
// two that leaves every lane group of the chip (8 per wave, 16 waves per CU) at most about one segment.
uint32_t choose_shift(const tcrc_ctx* ctx, uint64_t total) {
  const uint64_t groups = (uint64_t)ctx->cus * 16 * 8;
  uint32_t shift = 9;
  while (shift < 11 && (total >> shift) > 4096u) ++shift;  // up to 2 KiB while that keeps the launch within 512 rounds
  while (shift < 16 && (total >> shift) > groups) ++shift;
  return shift;
}

int shift_of(uint32_t segment_len, uint32_t* shift) {
  if (segment_len < 256u || (segment_len & (segment_len - 1u)) != 0u || segment_len > (1u << kPowMaxShift))
    return fail(TCRC_ERR_INVALID, "segment_len must be a power of two between 256 and 2^30");
  uint32_t s = 0;
  while ((1u << s) < segment_len) ++s;
  *shift = s;
  return TCRC_OK;
}

inline uint32_t segments_of(uint32_t len, uint32_t shift) {
  return len == 0 ? 1u : (uint32_t)(((uint64_t)len + (1ull << shift) - 1) >> shift);
}

}  // namespace

namespace tcrc_internal {

int long_device_setup(tcrc_ctx* ctx) {
  if (ctx->device < 0 || ctx->device >= kMaxDevices) return fail(TCRC_ERR_INVALID, "device index above 63");
  std::lock_guard<std::mutex> lock(g_static_mu);
  if (!g_static_ptr[ctx->device]) {
    // the array's name is only a handle on the host: this GPU's copy has an address of its own
    uint32_t* ptr = nullptr;
    HIP_TRY(hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch)));
    if (!ptr) return fail(TCRC_ERR_HIP, "hipGetSymbolAddress returned no address for the static scratch");
    // on a stream of its own, not the legacy default stream: work on that one would end a stream capture
    // another thread has open in global mode, and serialises against every blocking stream
    hipStream_t s = nullptr;
    HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipLaunchKernelGGL(k_zero_words, dim3(1), dim3(256), 0, s, ptr, kStaticAccWords);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) {
      hipLaunchKernelGGL(k_pow_tables, dim3(kPowTables), dim3(1024), 0, s, ptr + kStaticAccWords);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipStreamDestroy(s);
    HIP_TRY(e);
    g_static_ptr[ctx->device] = ptr;
  }
  // the least used slice of accumulator lines
  uint32_t best = 0;
  for (uint32_t i = 1; i < kStaticSlices; ++i)
    if (g_slice_users[ctx->device][i] < g_slice_users[ctx->device][best]) best = i;
  ++g_slice_users[ctx->device][best];
  ctx->static_slice = (int)best;
  ctx->d_static = g_static_ptr[ctx->device] + (size_t)best * kSliceWords;
  ctx->d_pow = g_static_ptr[ctx->device] + kStaticAccWords;
  return TCRC_OK;
}

void long_device_release(tcrc_ctx* ctx) {
  if (ctx->static_slice < 0 || ctx->device < 0 || ctx->device >= kMaxDevices) return;
  std::lock_guard<std::mutex> lock(g_static_mu);
  if (g_slice_users[ctx->device][ctx->static_slice]) --g_slice_users[ctx->device][ctx->static_slice];
  ctx->static_slice = -1;
}

}  // namespace tcrc_internal

namespace {

void fill_common(const tcrc_ctx* ctx, uint32_t shift, LongArgs* la) {
  la->seg_shift = shift;
  la->pow_lo = ctx->d_pow + (size_t)(shift - kPowMinShift) * kPowEntries;
  la->pow_hi = ctx->d_pow + (size_t)(shift + 11 - kPowMinShift) * kPowEntries;
}

// Validates the descriptors of a call with at most kLongInline buffers and puts them into the kernel
// arguments; returns the number of segments through *nseg.
int plan_inline(const tcrc_ctx* ctx, const tcrc_desc* h_desc, uint32_t n, uint32_t segment_len, LongArgs* la,
                uint64_t* nseg, uint64_t* total_bytes) {
  uint64_t total = 0;
  for (uint32_t i = 0; i < n; ++i) {
    if (h_desc[i].len > (1u << 30)) return fail(TCRC_ERR_TOO_LARGE, "buffer above 1 GiB: chain calls through init");
    total += h_desc[i].len;
  }
  uint32_t shift = 0;
  if (segment_len == 0) shift = choose_shift(ctx, total);
  else if (int rc = shift_of(segment_len, &shift)) return rc;
  memset(la, 0, sizeof(*la));
  fill_common(ctx, shift, la);
  la->nbuf = n;
  uint64_t count = 0;
  for (uint32_t i = 0; i < n; ++i) {
    la->inl_first[i] = (uint32_t)count;
    la->inl[i] = h_desc[i];
    count += segments_of(h_desc[i].len, shift);
  }
  if (count >= (1ull << 31)) return fail(TCRC_ERR_TOO_LARGE, "too many segments");
  *nseg = count;
  *total_bytes = total;
  return TCRC_OK;
}

// The context's accumulator lines (one per buffer of a launch) and, behind them, the line that holds the
// completion count; all zero between launches.
int acc_reserve(tcrc_ctx* ctx, hipStream_t s) {
  const size_t bytes = ((size_t)kLongLdsBuffers + 1) * 128;
  if (ctx->long_acc.bytes >= bytes) return TCRC_OK;
  if (int rc = scratch_reserve(ctx, &ctx->long_acc, bytes)) return rc;
  hipLaunchKernelGGL(k_zero_words, dim3(1), dim3(256), 0, s, (uint32_t*)ctx->long_acc.p, (uint32_t)(ctx->long_acc.bytes / 4));
  HIP_TRY(hipGetLastError());
  return TCRC_OK;
}

}  // namespace

extern "C" int tcrc_dev_extend_long(tcrc_ctx* ctx, const void* d_base, const tcrc_desc* h_desc, uint64_t n,
                                    uint32_t segment_len, uint32_t* d_out, void* stream) {
  if (!ctx) return fail(TCRC_ERR_INVALID, "null context");
  if (n == 0) return TCRC_OK;
  if (!d_base || !h_desc || !d_out) return fail(TCRC_ERR_INVALID, "null argument");
  if (n > (1u << 24)) return fail(TCRC_ERR_TOO_LARGE, "too many buffers in one long-buffer call");
  TCRC_DEVICE_SCOPE(ctx);
  hipStream_t s = (hipStream_t)stream;
  BatchParams p{};
  p.base = (const uint8_t*)d_base;
  p.out = d_out;

  if (n <= (uint64_t)kLongInline) {
    LongArgs la;
    uint64_t nseg = 0, total = 0;
    if (int rc = plan_inline(ctx, h_desc, (uint32_t)n, segment_len, &la, &nseg, &total)) return rc;
    std::lock_guard<std::mutex> lock(ctx->long_mu);  // the accumulator lines belong to one call at a time
    ctx->st_long_calls.fetch_add(1, std::memory_order_relaxed);
    ctx->st_long_bytes.fetch_add(total, std::memory_order_relaxed);
    if (int rc = acc_reserve(ctx, s)) return rc;
    la.acc = (uint32_t*)ctx->long_acc.p;
    p.n = nseg;
    // "returns with d_out complete", which also frees the lines for the next call: the kernel writes a word
    // of pinned memory when the last buffer's result is out, and the call watches that word
    if (ctx->poll_completion && !ctx->long_flag) {
      HIP_TRY(hipHostMalloc((void**)&ctx->long_flag, 64, hipHostMallocMapped | hipHostMallocCoherent));
      *ctx->long_flag = 0;  // the allocator hands back recently freed blocks as they were left
    }
    if (ctx->long_flag) {
      if (++ctx->long_seq == 0) ++ctx->long_seq;
      p.done.counter = (uint32_t*)ctx->long_acc.p + (size_t)kLongLdsBuffers * 32;  // the line behind the accumulators
      p.done.flag = ctx->long_flag;  // (mapped memory: one address for host and device)
      p.done.value = ctx->long_seq;
    }
    if (int rc = run_segments(ctx, p, la, stream)) return rc;
    HIP_TRY(ctx->long_flag ? wait_done(ctx->long_flag, ctx->long_seq, s) : hipStreamSynchronize(s));
    return TCRC_OK;
  }

  // More buffers than the kernel arguments hold: descriptors and running segment counts go to device memory,
  // and every launch takes up to kLongLdsBuffers buffers.
  uint64_t total = 0;
  for (uint64_t i = 0; i < n; ++i) {
    if (h_desc[i].len > (1u << 30)) return fail(TCRC_ERR_TOO_LARGE, "buffer above 1 GiB: chain calls through init");
    total += h_desc[i].len;
  }
  uint32_t shift = 0;
  if (segment_len == 0) shift = choose_shift(ctx, total);
  else if (int rc = shift_of(segment_len, &shift)) return rc;
  const size_t in_bytes = n * sizeof(tcrc_desc) + (n + 1) * 4;
  constexpr size_t kPinBytes = (size_t)kLongLdsBuffers * sizeof(tcrc_desc) + ((size_t)kLongLdsBuffers + 1) * 4;

  std::lock_guard<std::mutex> lock(ctx->long_mu);
  ctx->st_long_calls.fetch_add(1, std::memory_order_relaxed);
  ctx->st_long_bytes.fetch_add(total, std::memory_order_relaxed);
  // staging: the context's pinned block when it is large enough, else a pageable vector (the closing
  // synchronisation covers its copy)
  std::vector<uint32_t> pageable;
  uint8_t* stage;
  if (in_bytes <= kPinBytes) {
    if (!ctx->long_pin) HIP_TRY(hipHostMalloc((void**)&ctx->long_pin, kPinBytes, hipHostMallocDefault));
    stage = (uint8_t*)ctx->long_pin;
  } else {
    pageable.resize((in_bytes + 3) / 4);
    stage = (uint8_t*)pageable.data();
  }
  memcpy(stage, h_desc, n * sizeof(tcrc_desc));
  uint32_t* first = (uint32_t*)(stage + n * sizeof(tcrc_desc));
  uint64_t nseg = 0;
  for (uint64_t i = 0; i < n; ++i) {
    first[i] = (uint32_t)nseg;
    nseg += segments_of(h_desc[i].len, shift);
  }
  first[n] = (uint32_t)nseg;
  if (nseg >= (1ull << 31)) return fail(TCRC_ERR_TOO_LARGE, "too many segments");
  if (int rc = scratch_reserve(ctx, &ctx->long_in, in_bytes)) return rc;
  if (int rc = acc_reserve(ctx, s)) return rc;
  StreamGuard guard(s);  // the staging must not be left with a copy in flight
  HIP_TRY(hipMemcpyAsync(ctx->long_in.p, stage, in_bytes, hipMemcpyHostToDevice, s));
  const tcrc_desc* d_bufs = (const tcrc_desc*)ctx->long_in.p;
  const uint32_t* d_first = (const uint32_t*)((const uint8_t*)ctx->long_in.p + n * sizeof(tcrc_desc));
  for (uint64_t b0 = 0; b0 < n; b0 += kLongLdsBuffers) {
    const uint32_t nb = (uint32_t)(n - b0 < (uint64_t)kLongLdsBuffers ? n - b0 : (uint64_t)kLongLdsBuffers);
    LongArgs la;
    memset(&la, 0, sizeof(la));
    fill_common(ctx, shift, &la);
    la.nbuf = nb;
    la.first_base = first[b0];
    la.bufs = d_bufs + b0;
    la.seg_first = d_first + b0;
    la.acc = (uint32_t*)ctx->long_acc.p;  // launches of one stream run one after the other: the lines are free again
    p.n = first[b0 + nb] - first[b0];
    p.out = d_out + b0;
    if (int rc = run_segments(ctx, p, la, stream)) return rc;
  }
  guard.dismiss();
  HIP_TRY(hipStreamSynchronize(s));
  return TCRC_OK;
}

// tcrc_dev_extend_long without the wait: the kernel is enqueued on `stream` and the call returns; d_out is complete
// once the stream has run.  For callers with many such calls in a row (a backup engine checksumming file after
// file through its buffers): a call that waits pays the launch, the kernel's first microseconds - the table
// fill, one memory latency - and the wait for its last arrival, each time; enqueued back to back, one kernel's
// beginning runs under the end of the one before, and only the last call is waited for.
// At most kLongInline buffers (their descriptors travel in the kernel arguments; nothing is staged).  The
// accumulator lines belong to the context: the enqueued calls of one context must run one after the other - one
// stream, or events between streams.
extern "C" int tcrc_dev_extend_long_enqueue(tcrc_ctx* ctx, const void* d_base, const tcrc_desc* h_desc, uint64_t n,
                                            uint32_t segment_len, uint32_t* d_out, void* stream) {
  if (!ctx) return fail(TCRC_ERR_INVALID, "null context");
  if (n == 0) return TCRC_OK;
  if (!d_base || !h_desc || !d_out) return fail(TCRC_ERR_INVALID, "null argument");
  if (n > (uint64_t)kLongInline)
    return fail(TCRC_ERR_TOO_LARGE, "tcrc_dev_extend_long_enqueue takes at most 64 buffers per call");
  TCRC_DEVICE_SCOPE(ctx);
  LongArgs la;
  uint64_t nseg = 0, total = 0;
  if (int rc = plan_inline(ctx, h_desc, (uint32_t)n, segment_len, &la, &nseg, &total)) return rc;
  std::lock_guard<std::mutex> lock(ctx->long_mu);  // (for the duration of the enqueue: the lines' allocation)
  ctx->st_long_calls.fetch_add(1, std::memory_order_relaxed);
  ctx->st_long_bytes.fetch_add(total, std::memory_order_relaxed);
  if (int rc = acc_reserve(ctx, (hipStream_t)stream)) return rc;
  la.acc = (uint32_t*)ctx->long_acc.p;
  BatchParams p{};
  p.base = (const uint8_t*)d_base;
  p.out = d_out;
  p.n = nseg;
  return run_segments(ctx, p, la, stream);
}

// The same for callers that must not allocate (a HIP graph being captured, a serving loop, a harness that
// forbids it): at most kLongInline buffers, accumulator lines in the library's static device memory, one
// small kernel that clears the lines and the checksum kernel enqueued on `stream`, nothing else - no allocation,
// no copy command, no synchronisation.  Results are in d_out once the stream has run.  Calls that overlap on one GPU share the
// static lines and race: keep them on one stream, or order them with events.
extern "C" int tcrc_dev_extend_long_noalloc(tcrc_ctx* ctx, const void* d_base, const tcrc_desc* h_desc, uint64_t n,
                                            uint32_t segment_len, uint32_t* d_out, void* stream) {
  if (!ctx) return fail(TCRC_ERR_INVALID, "null context");
  if (n == 0) return TCRC_OK;
  if (!d_base || !h_desc || !d_out) return fail(TCRC_ERR_INVALID, "null argument");
  if (n > (uint64_t)kLongInline)
    return fail(TCRC_ERR_TOO_LARGE, "tcrc_dev_extend_long_noalloc takes at most 64 buffers per call");
  if (!ctx->d_static) return fail(TCRC_ERR_INVALID, "context without static device memory");
  TCRC_DEVICE_SCOPE(ctx);
  LongArgs la;
  uint64_t nseg = 0, total = 0;
  if (int rc = plan_inline(ctx, h_desc, (uint32_t)n, segment_len, &la, &nseg, &total)) return rc;
  ctx->st_long_calls.fetch_add(1, std::memory_order_relaxed);
  ctx->st_long_bytes.fetch_add(total, std::memory_order_relaxed);
  hipStream_t s = (hipStream_t)stream;
  // the lines keep whatever an earlier call (or one that was cut short) left in them
  if (ctx->debug_noalloc_memset) {
    // development aid (TCRC_DEBUG_NOALLOC_MEMSET=1): the form that went wrong under graph replay, kept so that
    // a trace of it can be taken
    HIP_TRY(hipMemsetAsync(ctx->d_static, 0, (size_t)kSliceWords * 4, s));
  } else {
    hipLaunchKernelGGL(k_zero_words, dim3(1), dim3(256), 0, s, ctx->d_static, kSliceWords);
    HIP_TRY(hipGetLastError());
  }
  la.acc = ctx->d_static;
  BatchParams p{};
  p.base = (const uint8_t*)d_base;
  p.out = d_out;
  p.n = nseg;
  return run_segments(ctx, p, la, stream);
}

// Development aid (not part of the C ABI in include/): copies the first `words` dwords of the context's slice of
// the static accumulator lines to the host, after waiting for the device.
extern "C" int tcrc_debug_static_lines(tcrc_ctx* ctx, uint32_t* h_out, size_t words) {
  if (!ctx || !h_out || !ctx->d_static) return fail(TCRC_ERR_INVALID, "null argument");
  if (words > kSliceWords) words = kSliceWords;
  TCRC_DEVICE_SCOPE(ctx);
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(h_out, ctx->d_static, words * 4, hipMemcpyDeviceToHost));
  return TCRC_OK;
}
