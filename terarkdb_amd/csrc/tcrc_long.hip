// Long buffers (hundreds of KiB to GiBs): the CRC of one buffer is computed by many lane
// groups at once.  A buffer is cut, from its END, into segments of S bytes (the first one
// may be shorter); every segment goes through the batch kernel as a buffer of its own, and
// the segment remainders are merged with the GF(2) identity
//     R(A || B) = R(A) * x^(8|B|)  xor  R(B)        (mod P)
// i.e. a segment followed by q whole segments is multiplied by x^(8 S q) and XORed into the
// buffer's result.  This is the role the reference's three-stream combine plays inside one
// call of crc32c_3way (util/crc32c.cc:641-658,686-699), widened from 3 streams to thousands;
// the caller-visible function is still crc32c::Extend (util/crc32c.cc:1226-1228), e.g. for
// the backup engine's whole-file checksums (utilities/backupable/backupable_db.cc:1434-1449).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string.h>

#include <string>
#include <vector>

#include "gf2.h"
#include "tcrc_internal.h"

using namespace tcrc_internal;

namespace {

struct ExpandParams {
  const tcrc_desc* buffers;    // [nbuf]
  const uint32_t* seg_first;   // [nbuf+1] segments before buffer b
  uint32_t nbuf;
  uint32_t nseg;
  uint32_t seg_len;
  tcrc_desc* segs;     // out [nseg]
  uint32_t* owner;     // out [nseg]
  uint32_t* after;     // out [nseg]
};

__global__ __launch_bounds__(256) void k_expand_segments(ExpandParams p) {
  const uint32_t s = blockIdx.x * 256u + threadIdx.x;
  if (s >= p.nseg) return;
  // buffer that owns segment s: last b with seg_first[b] <= s
  uint32_t lo = 0, hi = p.nbuf;
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (p.seg_first[mid] <= s) lo = mid; else hi = mid;
  }
  const uint32_t b = lo;
  const uint32_t k = s - p.seg_first[b];
  const uint32_t nseg = p.seg_first[b + 1] - p.seg_first[b];
  const tcrc_desc d = p.buffers[b];
  const uint32_t q = nseg - 1 - k;                    // whole segments after this one
  const uint64_t end = (uint64_t)d.len - (uint64_t)q * p.seg_len;
  const uint64_t start = end > p.seg_len ? end - p.seg_len : 0;
  tcrc_desc o;
  o.offset = d.offset + start;
  o.len = (uint32_t)(end - start);
  o.init = k == 0 ? d.init : 0xffffffffu;  // later segments start from a zero register
  p.segs[s] = o;
  p.owner[s] = b;
  p.after[s] = q;
}

// The kernel accumulates buffer b in acc[32 b] (one accumulator per 128-byte line).
__global__ __launch_bounds__(256) void k_long_gather(const uint32_t* acc, uint32_t* out, uint32_t n) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i < n) out[i] = acc[(uint64_t)i * 32u];
}

// seg_pow[q] = x^(8 * seg_len * q), grown on demand and cached per context.
int ensure_pow_table(tcrc_ctx* ctx, uint32_t seg_len, uint32_t count, hipStream_t s) {
  if (ctx->long_pow_seg == seg_len && ctx->long_pow_count >= count) return TCRC_OK;
  uint32_t want = count < 1024 ? 1024 : count + count / 2;
  std::vector<uint32_t> t(want);
  const uint32_t step = tcrc::gf2_xpow(8ull * seg_len);
  t[0] = 0x80000000u;
  for (uint32_t q = 1; q < want; ++q) t[q] = tcrc::gf2_mulmod(t[q - 1], step);
  if (int rc = scratch_reserve(ctx, &ctx->long_pow, (size_t)want * 4)) return rc;
  HIP_TRY(hipMemcpyAsync(ctx->long_pow.p, t.data(), (size_t)want * 4, hipMemcpyHostToDevice, s));
  HIP_TRY(hipStreamSynchronize(s));  // t goes out of scope
  ctx->long_pow_seg = seg_len;
  ctx->long_pow_count = want;
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
  uint64_t total = 0, longest = 0;
  for (uint64_t i = 0; i < n; ++i) {
    if (h_desc[i].len > (1u << 30)) return fail(TCRC_ERR_TOO_LARGE, "buffer above 1 GiB: chain calls through init");
    total += h_desc[i].len;
    if (h_desc[i].len > longest) longest = h_desc[i].len;
  }
  // Segment length: a power of two between 2 KiB and 64 KiB that gives every lane group of the
  // chip (8 per wave, 16 waves per CU) about one segment.
  uint32_t seg = segment_len;
  if (seg == 0) {
    const uint64_t groups = (uint64_t)ctx->cus * 16 * 8;
    seg = 2048;
    while (seg < 65536 && total / seg > groups) seg <<= 1;
  }
  if (seg < 256 || (seg & (seg - 1)) != 0) return fail(TCRC_ERR_INVALID, "segment_len must be a power of two >= 256");
  // one staging block: the descriptors, then the running segment counts
  std::vector<uint32_t> staging(n * 4 + n + 1);
  memcpy(staging.data(), h_desc, n * sizeof(tcrc_desc));
  uint32_t* first = staging.data() + n * 4;
  uint64_t nseg = 0;
  for (uint64_t i = 0; i < n; ++i) {
    first[i] = (uint32_t)nseg;
    const uint64_t k = h_desc[i].len == 0 ? 1 : (h_desc[i].len + seg - 1) / seg;
    nseg += k;
  }
  first[n] = (uint32_t)nseg;
  if (nseg >= (1ull << 31)) return fail(TCRC_ERR_TOO_LARGE, "too many segments");

  std::lock_guard<std::mutex> lock(ctx->long_mu);
  ctx->st_long_calls.fetch_add(1, std::memory_order_relaxed);
  ctx->st_long_bytes.fetch_add(total, std::memory_order_relaxed);
  if (int rc = ensure_pow_table(ctx, seg, (uint32_t)((longest + seg - 1) / seg + 1), s)) return rc;
  const size_t in_bytes = n * sizeof(tcrc_desc) + (n + 1) * 4;
  if (int rc = scratch_reserve(ctx, &ctx->long_in, in_bytes)) return rc;
  if (int rc = scratch_reserve(ctx, &ctx->long_desc, nseg * sizeof(tcrc_desc))) return rc;
  if (int rc = scratch_reserve(ctx, &ctx->long_owner, nseg * 4)) return rc;
  if (int rc = scratch_reserve(ctx, &ctx->long_after, nseg * 4)) return rc;
  uint8_t* in = (uint8_t*)ctx->long_in.p;
  HIP_TRY(hipMemcpyAsync(in, staging.data(), in_bytes, hipMemcpyHostToDevice, s));
  if (int rc = scratch_reserve(ctx, &ctx->long_acc, n * 128)) return rc;
  HIP_TRY(hipMemsetAsync(ctx->long_acc.p, 0, n * 128, s));
  ExpandParams e{};
  e.buffers = (const tcrc_desc*)in;
  e.seg_first = (const uint32_t*)(in + n * sizeof(tcrc_desc));
  e.nbuf = (uint32_t)n;
  e.nseg = (uint32_t)nseg;
  e.seg_len = seg;
  e.segs = (tcrc_desc*)ctx->long_desc.p;
  e.owner = (uint32_t*)ctx->long_owner.p;
  e.after = (uint32_t*)ctx->long_after.p;
  hipLaunchKernelGGL(k_expand_segments, dim3((unsigned)((nseg + 255) / 256)), dim3(256), 0, s, e);
  HIP_TRY(hipGetLastError());
  BatchParams p{};
  p.base = (const uint8_t*)d_base;
  p.desc = e.segs;
  p.n = nseg;
  p.uniform_suffix = -1;
  p.flags = kFlagSegments;
  p.lanes = 8;  // eight segments per wave are folded before one atomic: wider groups would queue on the accumulator
  p.out = (uint32_t*)ctx->long_acc.p;
  p.seg_owner = e.owner;
  p.seg_after = e.after;
  p.seg_pow = (const uint32_t*)ctx->long_pow.p;
  if (int rc = run_batch(ctx, p, stream)) return rc;
  hipLaunchKernelGGL(k_long_gather, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s,
                     (const uint32_t*)ctx->long_acc.p, d_out, (uint32_t)n);
  HIP_TRY(hipGetLastError());
  // the staging block was copied with pageable-memory semantics; make sure it is consumed
  HIP_TRY(hipStreamSynchronize(s));
  return TCRC_OK;
}
