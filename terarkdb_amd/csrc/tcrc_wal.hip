// Write-ahead-log images on the GPU: the host side of the scan (the kernel, k_wal_fused, sits with the
// CRC kernels in tcrc_engine.hip), the record list, and framing.
//
// A log file is a sequence of 32 KiB blocks; physical records never cross a block
// boundary (db/log_writer.cc:58-80), so every block can be parsed on its own.  The
// scan follows db/log_reader.cc:314-406 (ReadPhysicalRecord) for a complete image
// read without retries, in ONE kernel launch per 16 GiB: header walk (structural checks,
// db/log_reader.cc:317-384) and checksums (db/log_reader.cc:388-390) run side by side in every workgroup.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>

#include <string>

#include "tcrc_internal.h"

using namespace tcrc_internal;

namespace {

constexpr uint32_t kLogBlock = kLogBlockBytes;
constexpr uint32_t kHeader = kLogHeader;
constexpr uint32_t kRecyclableHeader = kLogRecyclableHeader;

// ---- record list -----------------------------------------------------------------------------

__global__ __launch_bounds__(256) void k_wal_block_counts(const tcrc_wal_block* blocks, uint64_t nblocks,
                                                          unsigned long long* counts) {
  const uint64_t b = (uint64_t)blockIdx.x * 256u + threadIdx.x;
  if (b <= nblocks) counts[b] = b < nblocks ? blocks[b].n_records : 0ull;  // one extra slot: the scan's total
}

struct ListParams {
  const uint8_t* img;
  uint64_t size;
  uint64_t log_number;
  uint64_t nblocks;
  const tcrc_wal_block* blocks;
  const unsigned long long* first;  // records before block b
  tcrc_wal_record* out;
  uint64_t capacity;
};

__global__ __launch_bounds__(256) void k_wal_list(ListParams p) {
  const uint64_t blk = (uint64_t)blockIdx.x * 256u + threadIdx.x;
  if (blk >= p.nblocks) return;
  const uint64_t base = blk * kLogBlock;
  const uint64_t end = base + kLogBlock < p.size ? base + kLogBlock : p.size;
  const bool last = end == p.size;
  const uint32_t n = p.blocks[blk].n_records, first_bad = p.blocks[blk].first_bad;
  uint64_t slot = p.first[blk], q = base;
  uint32_t hs = 0, len = 0, code = 0;
  for (uint32_t k = 0; k < n; ++k, ++slot) {
    if (!wal_step(p.img, q, end, last, p.log_number, &hs, &len, &code)) break;  // cannot happen for a matching d_blocks
    if (slot < p.capacity) {
      tcrc_wal_record r;
      r.offset = q;
      r.length = len;
      r.type = p.img[q + 6];
      r.status = k < first_bad ? TCRC_WAL_REC_OK : (k == first_bad ? TCRC_WAL_REC_BAD : TCRC_WAL_REC_SKIPPED);
      r.header_size = (uint16_t)hs;
      p.out[slot] = r;
    }
    q += hs + len;
  }
}

// ---- framing ---------------------------------------------------------------------------------

// One piece of the image: a physical record (header + payload copied from src) or, with type 0xff,
// a run of zero bytes (the tail of a block too short for a header).
struct Frag {
  uint64_t src;  // payload offset in the source
  uint64_t dst;  // offset of the header (or of the zero run) in the image
  uint32_t len;  // payload bytes (or zero bytes)
  uint32_t type;
};

struct FrameParams {
  const uint8_t* src;
  uint8_t* dst;
  const Frag* frags;
  uint32_t nfrags;
  uint32_t header_size;
  uint32_t log_number;
  tcrc_desc* spans;  // out: the checksummed span of every physical record (len 0 for zero runs)
};

// One wave per piece: header bytes, then the payload in 16-byte steps.
__global__ __launch_bounds__(256) void k_wal_frame(FrameParams p) {
  const uint32_t f = blockIdx.x * 4u + (threadIdx.x >> 6);
  if (f >= p.nfrags) return;
  const uint32_t lane = threadIdx.x & 63u;
  const Frag g = p.frags[f];
  uint8_t* out = p.dst + g.dst;
  if (g.type == 0xffu) {
    for (uint32_t i = lane; i < g.len; i += 64u) out[i] = 0;
    if (lane == 0) p.spans[f] = tcrc_desc{0, 0, 0};
    return;
  }
  const uint32_t hs = p.header_size;
  if (lane == 0) {
    out[0] = out[1] = out[2] = out[3] = 0;  // checksum: filled in by the CRC kernel
    out[4] = (uint8_t)(g.len & 0xffu);
    out[5] = (uint8_t)(g.len >> 8);
    out[6] = (uint8_t)g.type;
    if (hs == kRecyclableHeader) {
      out[7] = (uint8_t)p.log_number;
      out[8] = (uint8_t)(p.log_number >> 8);
      out[9] = (uint8_t)(p.log_number >> 16);
      out[10] = (uint8_t)(p.log_number >> 24);
    }
    tcrc_desc d;
    d.offset = g.dst + 6;
    d.len = g.len + hs - 6;
    d.init = 0;
    p.spans[f] = d;
  }
  const uint8_t* in = p.src + g.src;
  uint8_t* body = out + hs;
  const uint32_t whole = g.len & ~15u;
  for (uint32_t i = lane * 16u; i < whole; i += 1024u) {
    uint4 v;
    __builtin_memcpy(&v, in + i, 16);
    __builtin_memcpy(body + i, &v, 16);
  }
  if (lane < (g.len & 15u)) body[whole + lane] = in[whole + lane];
}

// The fragmentation rules of db/log_writer.cc:45-100 as a list of pieces; returns the image size.
uint64_t plan_frames(const tcrc_desc* recs, uint64_t n, bool recycle, uint32_t block_offset, std::vector<Frag>* out,
                     uint32_t* block_offset_out, uint64_t* physical) {
  const uint32_t hs = recycle ? kRecyclableHeader : kHeader;
  const uint32_t base_type = recycle ? 4u : 0u;  // recyclable types are the legacy ones + 4 (db/log_format.h:19-35)
  uint64_t pos = 0, phys = 0;
  uint32_t bo = block_offset;
  for (uint64_t i = 0; i < n; ++i) {
    uint64_t src = recs[i].offset;
    uint32_t left = recs[i].len;
    bool first = true;
    for (;;) {
      const uint32_t room = kLogBlock - bo;
      if (room < hs) {
        if (room > 0) {
          if (out) out->push_back(Frag{0, pos, room, 0xffu});
          pos += room;
        }
        bo = 0;
      }
      const uint32_t avail = kLogBlock - bo - hs;
      const uint32_t take = left < avail ? left : avail;
      const bool last = take == left;
      const uint32_t t = first ? (last ? 1u : 2u) : (last ? 4u : 3u);  // full, first, last, middle
      if (out) out->push_back(Frag{src, pos, take, t + base_type});
      pos += hs + take;
      bo += hs + take;
      src += take;
      left -= take;
      first = false;
      ++phys;
      if (left == 0) break;
    }
  }
  // an image that ends exactly at a block boundary continues at offset 0 of the next block (the writer's own
  // counter stands at 32768 then, db/log_writer.cc:58-80 turns it into 0 at the next record)
  if (block_offset_out) *block_offset_out = bo == kLogBlock ? 0u : bo;
  if (physical) *physical = phys;
  return pos;
}

}  // namespace

extern "C" uint64_t tcrc_wal_framed_size(const tcrc_desc* h_recs, uint64_t n, int recycle, uint32_t block_offset) {
  if (!h_recs || block_offset >= kLogBlock) return 0;
  return plan_frames(h_recs, n, recycle != 0, block_offset, nullptr, nullptr, nullptr);
}

extern "C" int tcrc_dev_wal_frame(tcrc_ctx* ctx, const void* d_src, const tcrc_desc* h_recs, uint64_t n,
                                  uint64_t log_number, int recycle, uint32_t block_offset, void* d_dst,
                                  uint64_t dst_capacity, uint64_t* h_written, uint32_t* block_offset_out,
                                  uint64_t* h_physical_records, void* stream) {
  if (!ctx) return fail(TCRC_ERR_INVALID, "null context");
  if (h_written) *h_written = 0;
  if (h_physical_records) *h_physical_records = 0;
  if (block_offset_out) *block_offset_out = block_offset;
  if (block_offset >= kLogBlock) return fail(TCRC_ERR_INVALID, "block_offset must be below 32768");
  if (n == 0) return TCRC_OK;
  if (!d_src || !h_recs || !d_dst) return fail(TCRC_ERR_INVALID, "null argument");
  std::vector<Frag> frags;
  uint32_t bo = 0;
  uint64_t phys = 0;
  const uint64_t total = plan_frames(h_recs, n, recycle != 0, block_offset, &frags, &bo, &phys);
  if (total > dst_capacity) return fail(TCRC_ERR_INVALID, "dst_capacity is smaller than the framed size");
  if (frags.size() >= (1ull << 32)) return fail(TCRC_ERR_TOO_LARGE, "too many physical records in one call");
  TCRC_DEVICE_SCOPE(ctx);
  std::lock_guard<std::mutex> lock(ctx->wal_mu);
  hipStream_t s = (hipStream_t)stream;
  const size_t frag_bytes = frags.size() * sizeof(Frag);
  const size_t span_off = (frag_bytes + 255) & ~(size_t)255;
  if (int rc = scratch_reserve(ctx, &ctx->wal_frags, span_off + frags.size() * sizeof(tcrc_desc))) return rc;
  StreamGuard guard(s);  // `frags` must not go away with its copy in flight
  HIP_TRY(hipMemcpyAsync(ctx->wal_frags.p, frags.data(), frag_bytes, hipMemcpyHostToDevice, s));
  FrameParams f{};
  f.src = (const uint8_t*)d_src;
  f.dst = (uint8_t*)d_dst;
  f.frags = (const Frag*)ctx->wal_frags.p;
  f.nfrags = (uint32_t)frags.size();
  f.header_size = recycle ? kRecyclableHeader : kHeader;
  f.log_number = (uint32_t)log_number;
  f.spans = (tcrc_desc*)((uint8_t*)ctx->wal_frags.p + span_off);
  hipLaunchKernelGGL(k_wal_frame, dim3((f.nfrags + 3u) / 4u), dim3(256), 0, s, f);
  HIP_TRY(hipGetLastError());
  BatchParams p{};
  p.base = f.dst;
  p.desc = f.spans;
  p.n = frags.size();
  p.uniform_suffix = -1;
  p.flags = TCRC_F_MASK | kFlagStampBefore;  // zero runs have empty spans, which the kernel does not stamp
  if (int rc = run_batch(ctx, p, stream)) return rc;
  guard.dismiss();
  HIP_TRY(hipStreamSynchronize(s));  // the piece list was copied from pageable memory
  if (h_written) *h_written = total;
  if (block_offset_out) *block_offset_out = bo;
  if (h_physical_records) *h_physical_records = phys;
  return TCRC_OK;
}

extern "C" int tcrc_dev_wal_list(tcrc_ctx* ctx, const void* d_img, uint64_t size, uint64_t log_number,
                                 const tcrc_wal_block* d_blocks, tcrc_wal_record* d_records, uint64_t capacity,
                                 uint64_t* h_count, void* stream) {
  if (!ctx) return fail(TCRC_ERR_INVALID, "null context");
  if (h_count) *h_count = 0;
  if (size == 0) return TCRC_OK;
  if (!d_img || !d_blocks || (capacity && !d_records)) return fail(TCRC_ERR_INVALID, "null argument");
  TCRC_DEVICE_SCOPE(ctx);
  std::lock_guard<std::mutex> lock(ctx->wal_mu);
  hipStream_t s = (hipStream_t)stream;
  const uint64_t nblocks = (size + kLogBlock - 1) / kLogBlock;
  if (nblocks >= (1ull << 31)) return fail(TCRC_ERR_TOO_LARGE, "log image above 64 TiB");
  const size_t words = (size_t)nblocks + 1;
  if (int rc = scratch_reserve(ctx, &ctx->wal_list_counts, words * 8)) return rc;
  if (int rc = scratch_reserve(ctx, &ctx->wal_list_first, words * 8)) return rc;
  unsigned long long* counts = (unsigned long long*)ctx->wal_list_counts.p;
  unsigned long long* first = (unsigned long long*)ctx->wal_list_first.p;
  hipLaunchKernelGGL(k_wal_block_counts, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, s, d_blocks, nblocks,
                     counts);
  HIP_TRY(hipGetLastError());
  size_t temp_bytes = 0;
  HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, temp_bytes, counts, first, (int)words, s));
  if (int rc = scratch_reserve(ctx, &ctx->wal_list_temp, temp_bytes + 256)) return rc;
  HIP_TRY(hipcub::DeviceScan::ExclusiveSum(ctx->wal_list_temp.p, temp_bytes, counts, first, (int)words, s));
  ListParams p{};
  p.img = (const uint8_t*)d_img;
  p.size = size;
  p.log_number = log_number;
  p.nblocks = nblocks;
  p.blocks = d_blocks;
  p.first = first;
  p.out = d_records;
  p.capacity = capacity;
  if (capacity) {
    hipLaunchKernelGGL(k_wal_list, dim3((unsigned)((nblocks + 255) / 256)), dim3(256), 0, s, p);
    HIP_TRY(hipGetLastError());
  }
  unsigned long long total = 0;
  HIP_TRY(hipMemcpyAsync(&total, first + nblocks, 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  if (h_count) *h_count = total;
  return TCRC_OK;
}

extern "C" int tcrc_dev_wal_verify(tcrc_ctx* ctx, const void* d_img, uint64_t size, uint64_t log_number,
                                   tcrc_wal_block* d_blocks, uint64_t* d_counts, uint64_t* h_records,
                                   void* stream) {
  return tcrc_dev_wal_verify_part(ctx, d_img, size, log_number, 0, 0, d_blocks, d_counts, h_records, stream);
}

extern "C" int tcrc_dev_wal_verify_part(tcrc_ctx* ctx, const void* d_img, uint64_t size, uint64_t log_number,
                                        uint64_t first_block, int more_follows, tcrc_wal_block* d_blocks,
                                        uint64_t* d_counts, uint64_t* h_records, void* stream) {
  if (!ctx) return fail(TCRC_ERR_INVALID, "null context");
  if (h_records) *h_records = 0;
  if (more_follows && size % kLogBlock != 0)
    return fail(TCRC_ERR_INVALID, "a part that is not the end of the log must be whole 32 KiB blocks");
  if (size == 0) return TCRC_OK;
  if (!d_img || !d_blocks || !d_counts) return fail(TCRC_ERR_INVALID, "null argument");
  TCRC_DEVICE_SCOPE(ctx);
  ctx->st_log_images.fetch_add(1, std::memory_order_relaxed);
  ctx->st_log_bytes.fetch_add(size, std::memory_order_relaxed);
  hipStream_t s = (hipStream_t)stream;
  const uint64_t nblocks_total = (size + kLogBlock - 1) / kLogBlock;
  // The record count goes back to the host only if it is asked for, and then without a copy command: the
  // kernels add it into one of 32 words of pinned host memory (a slot per call, so that a scan still running
  // on another stream does not disturb it) through two device words of the same slot.
  unsigned long long *d_words = nullptr, *pin_records = nullptr;
  if (h_records) {
    std::lock_guard<std::mutex> lock(ctx->wal_mu);
    if (!ctx->wal_misc.p) {
      if (int rc = scratch_reserve(ctx, &ctx->wal_misc, 64 * sizeof(unsigned long long))) return rc;
      HIP_TRY(hipMemset(ctx->wal_misc.p, 0, ctx->wal_misc.bytes));
    }
    if (!ctx->wal_pin)
      HIP_TRY(hipHostMalloc((void**)&ctx->wal_pin, 32 * sizeof(unsigned long long), hipHostMallocMapped | hipHostMallocCoherent));
    const uint64_t slot = ctx->wal_calls++ % 32;
    d_words = (unsigned long long*)ctx->wal_misc.p + 2 * slot;
    pin_records = ctx->wal_pin + slot;
    *pin_records = 0;
  }
  for (uint64_t b0 = 0; b0 < nblocks_total; b0 += kWalMaxBlocks) {
    const uint32_t nb = (uint32_t)(nblocks_total - b0 < kWalMaxBlocks ? nblocks_total - b0 : kWalMaxBlocks);
    const uint64_t off = b0 * kLogBlock;
    WalScanParams w{};
    w.img = (const uint8_t*)d_img + off;
    w.size = size - off < (uint64_t)nb * kLogBlock ? size - off : (uint64_t)nb * kLogBlock;
    w.log_number = log_number;
    w.block_base = first_block + b0;
    w.nblocks = nb;
    w.tail = (!more_follows && b0 + nb == nblocks_total) ? 1u : 0u;  // only the file's last block sees its end
    w.blocks = d_blocks + b0;
    w.counts = (unsigned long long*)d_counts;
    w.records_host = pin_records;
    w.launch_words = d_words;
    if (int rc = run_wal_scan(ctx, w, stream)) return rc;
  }
  uint64_t total_records = 0;
  if (h_records) {
    HIP_TRY(hipStreamSynchronize(s));
    total_records = *(volatile unsigned long long*)pin_records;
  }
  if (h_records) *h_records = total_records;
  return TCRC_OK;
}

// Development aid (not part of the C ABI in include/): scans of this context write per-workgroup phase
// timestamps (32 x uint64 per CU, 100 MHz ticks) into d_trace until it is set back to NULL.
extern "C" int tcrc_debug_wal_trace(tcrc_ctx* ctx, void* d_trace) {
  if (!ctx) return fail(TCRC_ERR_INVALID, "null context");
  ctx->wal_trace = d_trace;
  return TCRC_OK;
}
