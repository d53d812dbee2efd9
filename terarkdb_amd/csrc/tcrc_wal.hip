// Write-ahead-log scanning on the GPU.
//
// A log file is a sequence of 32 KiB blocks; physical records never cross a block
// boundary (db/log_writer.cc:58-80), so every block can be parsed on its own.  The
// scan follows db/log_reader.cc:314-406 (ReadPhysicalRecord) for a complete image
// read without retries:
//   1. k_wal_walk<false>: one thread per block walks the headers and counts the
//      records that pass every structural check (header fits, log number matches,
//      length fits in the block, not a zero-length zero-type filler);
//   2. k_scan_counts: exclusive prefix sum of the counts;
//   3. k_wal_walk<true>: the same walk, now writing one descriptor per record: the
//      CRC-covered span starts at header+6 (type byte) and runs header_size-6+length
//      bytes (db/log_reader.cc:389);
//   4. the batch CRC kernel checks every span against Unmask(fixed32(header))
//      (db/log_reader.cc:388-390) and keeps, per block, the smallest failing record
//      index: the reference drops the rest of a block at its first bad checksum
//      (db/log_reader.cc:391-397), so that index is all the policy code needs.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "tcrc_internal.h"

using namespace tcrc_internal;

namespace {

constexpr uint32_t kLogBlock = 32768;  // db/log_format.h:37
constexpr uint32_t kHeader = 7;        // db/log_format.h:40
constexpr uint32_t kRecyclableHeader = 11;  // db/log_format.h:44

struct WalkParams {
  const uint8_t* img;
  uint64_t size;        // bytes in this image (the last block may be short)
  uint64_t log_number;
  uint32_t nblocks;
  uint32_t* count;      // [nblocks]   (count pass: out, emit pass: unused)
  const uint32_t* first;  // [nblocks+1] exclusive prefix of count
  tcrc_desc* items;
  uint32_t* item_block;
  uint64_t capacity;
  tcrc_wal_block* blocks;  // [nblocks] emit pass: n_records / first_bad / end_offset / end_code
};

template <bool EMIT>
__global__ __launch_bounds__(256) void k_wal_walk(WalkParams p) {
  const uint32_t blk = blockIdx.x * 256u + threadIdx.x;
  if (blk >= p.nblocks) return;
  const uint64_t base = (uint64_t)blk * kLogBlock;
  const uint64_t end = base + kLogBlock < p.size ? base + kLogBlock : p.size;
  const bool last = end == p.size;
  uint64_t pos = base;
  uint32_t n = 0;
  uint32_t code = TCRC_WAL_END_CLEAN;
  const uint64_t slot0 = EMIT ? p.first[blk] : 0;
  for (;;) {
    const uint64_t remaining = end - pos;
    if (remaining < kHeader) {
      // a block trailer (zero padding), or at the very end of the file a truncated header
      code = remaining == 0 ? TCRC_WAL_END_CLEAN : (last ? TCRC_WAL_END_SHORT_AT_EOF : TCRC_WAL_END_TRAILER);
      break;
    }
    const uint8_t* h = p.img + pos;
    const uint32_t len = (uint32_t)h[4] | ((uint32_t)h[5] << 8);
    const uint32_t type = (uint32_t)(int)(signed char)h[6];  // the reference reads it through a char
    uint32_t hs = kHeader;
    if (type >= 5u && type <= 8u) {
      hs = kRecyclableHeader;
      if (remaining < kRecyclableHeader) {
        code = last ? TCRC_WAL_END_SHORT_AT_EOF : TCRC_WAL_END_TRAILER;
        break;
      }
      const uint32_t lognum = (uint32_t)h[7] | ((uint32_t)h[8] << 8) | ((uint32_t)h[9] << 16) | ((uint32_t)h[10] << 24);
      if ((uint64_t)lognum != p.log_number) {
        code = TCRC_WAL_END_OLD_RECORD;
        break;
      }
    }
    if ((uint64_t)hs + len > remaining) {
      code = last ? TCRC_WAL_END_BAD_LEN_AT_EOF : TCRC_WAL_END_BAD_LEN;
      break;
    }
    if (type == 0u && len == 0u) {
      code = TCRC_WAL_END_ZERO_RECORD;
      break;
    }
    if (EMIT) {
      const uint64_t slot = slot0 + n;
      if (slot < p.capacity) {
        tcrc_desc d;
        d.offset = pos + 6;
        d.len = len + hs - 6;
        d.init = n;
        p.items[slot] = d;
        p.item_block[slot] = blk;
      }
    }
    ++n;
    pos += hs + len;
  }
  if (EMIT) {
    tcrc_wal_block r;
    r.n_records = n;
    r.first_bad = 0xffffffffu;
    r.end_offset = (uint32_t)(pos - base);
    r.end_code = code;
    p.blocks[blk] = r;
  } else {
    p.count[blk] = n;
  }
}

// out[i] = sum(in[0..i)), out[n] = total; one workgroup of 16 waves.  1024 elements per pass:
// a shuffle scan inside each wave, the 16 wave totals through LDS, a running carry across passes.
__global__ __launch_bounds__(1024) void k_scan_counts(const uint32_t* in, uint32_t* out, uint32_t n,
                                                      unsigned long long* total) {
  __shared__ uint32_t wave_sum[16];
  const uint32_t t = threadIdx.x, lane = t & 63u, w = t >> 6;
  uint32_t carry = 0;
  for (uint32_t base = 0; base < n; base += 1024u) {
    const uint32_t i = base + t;
    const uint32_t v = i < n ? in[i] : 0u;
    uint32_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t up = __shfl_up(incl, d);
      if ((int)lane >= d) incl += up;
    }
    if (lane == 63u) wave_sum[w] = incl;
    __syncthreads();
    uint32_t before = 0, all = 0;
#pragma unroll
    for (uint32_t k = 0; k < 16u; ++k) {
      const uint32_t sk = wave_sum[k];
      before += k < w ? sk : 0u;
      all += sk;
    }
    if (i < n) out[i] = carry + before + incl - v;
    carry += all;
    __syncthreads();
  }
  if (t == 0) {
    out[n] = carry;
    *total = carry;
  }
}

}  // namespace

extern "C" int tcrc_dev_wal_verify(tcrc_ctx* ctx, const void* d_img, uint64_t size, uint64_t log_number,
                                   tcrc_wal_block* d_blocks, uint64_t* d_counts, uint64_t* h_records,
                                   void* stream) {
  if (!ctx) return fail(TCRC_ERR_INVALID, "null context");
  if (h_records) *h_records = 0;
  if (size == 0) return TCRC_OK;
  if (!d_img || !d_blocks || !d_counts) return fail(TCRC_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t s = (hipStream_t)stream;
  const uint64_t nblocks_total = (size + kLogBlock - 1) / kLogBlock;
  // One slice is up to 8 GiB (the walk kernels are latency-bound, so fewer, larger launches are
  // cheaper).  Each slice needs one host round trip: the record count sizes the descriptor list.
  const uint64_t slice_blocks = 262144;
  uint64_t total_records = 0;
  for (uint64_t b0 = 0; b0 < nblocks_total; b0 += slice_blocks) {
    const uint32_t nb = (uint32_t)(nblocks_total - b0 < slice_blocks ? nblocks_total - b0 : slice_blocks);
    const uint64_t off = b0 * kLogBlock;
    const uint64_t bytes = size - off < (uint64_t)nb * kLogBlock ? size - off : (uint64_t)nb * kLogBlock;
    if (int rc = scratch_reserve(ctx, &ctx->wal_counts, (size_t)nb * 4)) return rc;
    if (int rc = scratch_reserve(ctx, &ctx->wal_first, ((size_t)nb + 1) * 4)) return rc;
    if (int rc = scratch_reserve(ctx, &ctx->wal_misc, 64)) return rc;
    unsigned long long* d_total = (unsigned long long*)ctx->wal_misc.p;
    WalkParams w{};
    w.img = (const uint8_t*)d_img + off;
    w.size = bytes;
    w.log_number = log_number;
    w.nblocks = nb;
    w.count = (uint32_t*)ctx->wal_counts.p;
    w.first = (const uint32_t*)ctx->wal_first.p;
    w.blocks = d_blocks + b0;
    const unsigned grid = (nb + 255u) / 256u;
    hipLaunchKernelGGL(k_wal_walk<false>, dim3(grid), dim3(256), 0, s, w);
    hipLaunchKernelGGL(k_scan_counts, dim3(1), dim3(1024), 0, s, (const uint32_t*)ctx->wal_counts.p,
                       (uint32_t*)ctx->wal_first.p, nb, d_total);
    HIP_TRY(hipGetLastError());
    unsigned long long total = 0;
    HIP_TRY(hipMemcpyAsync(&total, d_total, 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    const uint64_t capacity = total + 1;
    if (int rc = scratch_reserve(ctx, &ctx->wal_items, capacity * sizeof(tcrc_desc))) return rc;
    if (int rc = scratch_reserve(ctx, &ctx->wal_item_block, capacity * 4)) return rc;
    w.items = (tcrc_desc*)ctx->wal_items.p;
    w.item_block = (uint32_t*)ctx->wal_item_block.p;
    w.capacity = capacity;
    hipLaunchKernelGGL(k_wal_walk<true>, dim3(grid), dim3(256), 0, s, w);
    HIP_TRY(hipGetLastError());
    if (total) {
      BatchParams p{};
      p.base = w.img;
      p.desc = w.items;
      p.n = total;
      p.uniform_suffix = -1;
      p.flags = TCRC_F_MASK | TCRC_F_EXPECT_BEFORE | kFlagWalItems;
      p.verify = 1;
      p.counts = (unsigned long long*)d_counts;
      p.item_block = w.item_block;
      p.wal_blocks = w.blocks;
      if (int rc = run_batch(ctx, p, stream)) return rc;
    }
    total_records += total;
  }
  if (h_records) *h_records = total_records;
  return TCRC_OK;
}
