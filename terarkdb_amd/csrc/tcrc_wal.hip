// Write-ahead-log scanning on the GPU.
//
// A log file is a sequence of 32 KiB blocks; physical records never cross a block
// boundary (db/log_writer.cc:58-80), so every block can be parsed on its own.  The
// scan follows db/log_reader.cc:314-406 (ReadPhysicalRecord) for a complete image
// read without retries:
//   1. k_wal_walk: one thread per block walks the headers and counts the records that
//      pass every structural check (header fits, log number matches, length fits in
//      the block, not a zero-length zero-type filler), takes descriptor slots for them
//      and walks again writing one descriptor per record: the CRC-covered span starts
//      at header+6 (type byte) and runs header_size-6+length bytes
//      (db/log_reader.cc:389).  Records of 4 KiB and more go to a second list;
//   2. the batch CRC kernel checks every span against Unmask(fixed32(header))
//      (db/log_reader.cc:388-390), four lanes per short record and a whole wave per
//      long one, and keeps, per block, the smallest failing record index: the
//      reference drops the rest of a block at its first bad checksum
//      (db/log_reader.cc:391-397), so that index is all the policy code needs.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>

#include <string>

#include "tcrc_internal.h"

using namespace tcrc_internal;

namespace {

constexpr uint32_t kLogBlock = 32768;  // db/log_format.h:37
constexpr uint32_t kHeader = 7;        // db/log_format.h:40
constexpr uint32_t kRecyclableHeader = 11;  // db/log_format.h:44
// Spans at least this long are checked by a whole wave each: with four lanes a 32 KiB fragment
// takes ~160 us, which a log of large records (few, long spans) cannot hide behind other work.
constexpr uint32_t kLongSpan = 4096;

struct WalkParams {
  const uint8_t* img;
  uint64_t size;        // bytes in this image (the last block may be short)
  uint32_t tail;        // the image ends where the file ends (0: more of the file follows this part)
  uint64_t log_number;
  uint32_t nblocks;
  tcrc_desc* items;  // short spans
  uint32_t* item_block;
  uint64_t capacity;
  tcrc_desc* items_long;  // spans of kLongSpan bytes and more
  uint32_t* item_block_long;
  uint64_t capacity_long;
  tcrc_wal_block* blocks;       // [nblocks] n_records / first_bad / end_offset / end_code
  unsigned long long* n_items;  // [2] running totals of short / long descriptors handed out
};

// One header step of db/log_reader.cc:317-384.  Returns false when the walk of the block ends
// (code says why); otherwise the record occupies [pos, pos + hs + len).
__device__ __forceinline__ bool wal_step(const uint8_t* img, uint64_t pos, uint64_t end, bool last,
                                         uint64_t log_number, uint32_t* hs_out, uint32_t* len_out,
                                         uint32_t* code) {
  const uint64_t remaining = end - pos;
  if (remaining < kHeader) {
    // a block trailer (zero padding), or at the very end of the file a truncated header
    *code = remaining == 0 ? TCRC_WAL_END_CLEAN : (last ? TCRC_WAL_END_SHORT_AT_EOF : TCRC_WAL_END_TRAILER);
    return false;
  }
  const uint8_t* h = img + pos;
  const uint32_t len = (uint32_t)h[4] | ((uint32_t)h[5] << 8);
  const uint32_t type = (uint32_t)(int)(signed char)h[6];  // the reference reads it through a char
  uint32_t hs = kHeader;
  if (type >= 5u && type <= 8u) {
    hs = kRecyclableHeader;
    if (remaining < kRecyclableHeader) {
      *code = last ? TCRC_WAL_END_SHORT_AT_EOF : TCRC_WAL_END_TRAILER;
      return false;
    }
    const uint32_t lognum = (uint32_t)h[7] | ((uint32_t)h[8] << 8) | ((uint32_t)h[9] << 16) | ((uint32_t)h[10] << 24);
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

__device__ __forceinline__ uint32_t wave_exclusive_sum(uint32_t v, uint32_t lane, uint32_t* total) {
  uint32_t incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t up = __shfl_up(incl, d);
    if ((int)lane >= d) incl += up;
  }
  *total = __shfl(incl, 63);
  return incl - v;
}

// One thread per 32 KiB block.  Pass 1 counts the block's short and long records; the 64 counts of a
// wave are prefix-summed with shuffles and the wave takes its runs of descriptor slots with ONE atomic
// add per list; pass 2 writes the descriptors from the header facts pass 1 kept in LDS (it walks memory
// again only past the 32nd record of a block).  Descriptors
// of one block are contiguous within a list, blocks land in no particular order - nothing downstream
// depends on it.
__global__ __launch_bounds__(256) void k_wal_walk(WalkParams p) {
  // (header position, header size, payload length) of the first kCached records of every block, kept
  // from the counting pass so that the emitting pass does not chase the headers through memory again
  constexpr uint32_t kCached = 32;
  __shared__ uint32_t cache[kCached * 256];
  const uint32_t blk = blockIdx.x * 256u + threadIdx.x;
  const bool have = blk < p.nblocks;
  const uint64_t base = (uint64_t)blk * kLogBlock;
  const uint64_t end = base + kLogBlock < p.size ? base + kLogBlock : p.size;
  const bool last = p.tail && end == p.size;
  uint32_t n = 0, n_long = 0, code = TCRC_WAL_END_CLEAN, hs = 0, len = 0;
  uint64_t pos = base;
  if (have) {
    while (wal_step(p.img, pos, end, last, p.log_number, &hs, &len, &code)) {
      // 15 bits of position, 1 bit of header kind, 16 bits of length
      if (n < kCached)
        cache[n * 256u + threadIdx.x] = (uint32_t)(pos - base) | (hs == kRecyclableHeader ? 0x8000u : 0u) | (len << 16);
      ++n;
      if (len + hs - 6 >= kLongSpan) ++n_long;
      pos += hs + len;
    }
  }
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t total_s, total_l;
  const uint32_t excl_s = wave_exclusive_sum(n - n_long, lane, &total_s);
  const uint32_t excl_l = wave_exclusive_sum(n_long, lane, &total_l);
  unsigned long long base_s = 0, base_l = 0;
  if (lane == 63u && total_s) base_s = atomicAdd(&p.n_items[0], (unsigned long long)total_s);
  if (lane == 63u && total_l) base_l = atomicAdd(&p.n_items[1], (unsigned long long)total_l);
  base_s = __shfl(base_s, 63);
  base_l = __shfl(base_l, 63);
  if (!have) return;
  uint64_t slot_s = base_s + excl_s, slot_l = base_l + excl_l;
  uint64_t q = base;
  for (uint32_t k = 0; k < n; ++k) {
    if (k < kCached) {
      const uint32_t c = cache[k * 256u + threadIdx.x];
      q = base + (c & 0x7fffu);
      hs = (c & 0x8000u) ? kRecyclableHeader : kHeader;
      len = c >> 16;
    } else {
      uint32_t c2;
      wal_step(p.img, q, end, last, p.log_number, &hs, &len, &c2);
    }
    tcrc_desc d;
    d.offset = q + 6;  // the CRC covers the type byte, the log number if any, and the payload
    d.len = len + hs - 6;
    d.init = k;
    if (d.len >= kLongSpan) {
      if (slot_l < p.capacity_long) {
        p.items_long[slot_l] = d;
        p.item_block_long[slot_l] = blk;
      }
      ++slot_l;
    } else {
      if (slot_s < p.capacity) {
        p.items[slot_s] = d;
        p.item_block[slot_s] = blk;
      }
      ++slot_s;
    }
    q += hs + len;
  }
  tcrc_wal_block r;
  r.n_records = n;
  r.first_bad = 0xffffffffu;
  r.end_offset = (uint32_t)(pos - base);
  r.end_code = code;
  p.blocks[blk] = r;
}

// misc: [0] short descriptors wanted, [1] long ones, [2..4] TCRC_CNT_* of this slice.  The slice's counters
// reach the caller's block only if every descriptor fitted (otherwise the slice is redone).
__global__ void k_wal_init(unsigned long long* misc) {
  misc[0] = 0ull;
  misc[1] = 0ull;
  misc[2 + TCRC_CNT_MISMATCH] = 0ull;
  misc[2 + TCRC_CNT_FIRST_BAD] = ~0ull;
  misc[2 + TCRC_CNT_CHECKED] = 0ull;
}

__global__ void k_wal_finish(const unsigned long long* misc, unsigned long long capacity,
                             unsigned long long capacity_long, unsigned long long* counts) {
  if (misc[0] > capacity || misc[1] > capacity_long) return;
  atomicAdd(&counts[TCRC_CNT_MISMATCH], misc[2 + TCRC_CNT_MISMATCH]);
  atomicMin(&counts[TCRC_CNT_FIRST_BAD], misc[2 + TCRC_CNT_FIRST_BAD]);
  atomicAdd(&counts[TCRC_CNT_CHECKED], misc[2 + TCRC_CNT_CHECKED]);
}

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
  if (block_offset_out) *block_offset_out = bo;
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
  std::lock_guard<std::mutex> lock(ctx->wal_mu);  // the descriptor scratch belongs to one scan at a time
  ctx->st_log_images.fetch_add(1, std::memory_order_relaxed);
  ctx->st_log_bytes.fetch_add(size, std::memory_order_relaxed);
  hipStream_t s = (hipStream_t)stream;
  const uint64_t nblocks_total = (size + kLogBlock - 1) / kLogBlock;
  // One slice is up to 8 GiB (the walk is latency-bound, so fewer, larger launches are cheaper).
  const uint64_t slice_blocks = 262144;
  uint64_t total_records = 0;
  for (uint64_t b0 = 0; b0 < nblocks_total; b0 += slice_blocks) {
    const uint32_t nb = (uint32_t)(nblocks_total - b0 < slice_blocks ? nblocks_total - b0 : slice_blocks);
    const uint64_t off = b0 * kLogBlock;
    const uint64_t bytes = size - off < (uint64_t)nb * kLogBlock ? size - off : (uint64_t)nb * kLogBlock;
    // Descriptor room for one short record per 32 bytes and one long one per kLongSpan bytes; a log
    // made of smaller records is redone once with the exact counts the first attempt reports.
    uint64_t capacity = bytes / 32 + 1024, capacity_long = bytes / kLongSpan + 1024;
    for (int attempt = 0;; ++attempt) {
      const size_t long_off = (capacity * sizeof(tcrc_desc) + 255) & ~(size_t)255;
      const size_t long_off_b = (capacity * 4 + 255) & ~(size_t)255;
      if (int rc = scratch_reserve(ctx, &ctx->wal_items, long_off + capacity_long * sizeof(tcrc_desc))) return rc;
      if (int rc = scratch_reserve(ctx, &ctx->wal_item_block, long_off_b + capacity_long * 4)) return rc;
      if (int rc = scratch_reserve(ctx, &ctx->wal_misc, 64)) return rc;
      unsigned long long* misc = (unsigned long long*)ctx->wal_misc.p;
      WalkParams w{};
      w.img = (const uint8_t*)d_img + off;
      w.size = bytes;
      w.tail = (!more_follows && b0 + nb == nblocks_total) ? 1u : 0u;  // only the file's last block sees its end
      w.log_number = log_number;
      w.nblocks = nb;
      w.items = (tcrc_desc*)ctx->wal_items.p;
      w.item_block = (uint32_t*)ctx->wal_item_block.p;
      w.capacity = capacity;
      w.items_long = (tcrc_desc*)((uint8_t*)ctx->wal_items.p + long_off);
      w.item_block_long = (uint32_t*)((uint8_t*)ctx->wal_item_block.p + long_off_b);
      w.capacity_long = capacity_long;
      w.blocks = d_blocks + b0;
      w.n_items = misc;
      hipLaunchKernelGGL(k_wal_init, dim3(1), dim3(1), 0, s, misc);
      hipLaunchKernelGGL(k_wal_walk, dim3((nb + 255u) / 256u), dim3(256), 0, s, w);
      HIP_TRY(hipGetLastError());
      BatchParams p{};
      p.base = w.img;
      p.uniform_suffix = -1;
      p.flags = TCRC_F_MASK | TCRC_F_EXPECT_BEFORE | kFlagWalItems;
      p.verify = 1;
      p.counts = misc + 2;
      p.wal_blocks = w.blocks;
      p.wal_block_base = first_block + b0;
      // short spans: four lanes each; the kernel takes min(capacity, descriptors written)
      p.desc = w.items;
      p.n = capacity;
      p.n_dev = misc;
      p.item_block = w.item_block;
      p.lanes = 4;
      if (int rc = run_batch(ctx, p, stream)) return rc;
      // long spans: a wave each
      p.desc = w.items_long;
      p.n = capacity_long;
      p.n_dev = misc + 1;
      p.item_block = w.item_block_long;
      p.lanes = 64;
      if (int rc = run_batch(ctx, p, stream)) return rc;
      hipLaunchKernelGGL(k_wal_finish, dim3(1), dim3(1), 0, s, misc, (unsigned long long)capacity,
                         (unsigned long long)capacity_long, (unsigned long long*)d_counts);
      HIP_TRY(hipGetLastError());
      unsigned long long wanted[2] = {0, 0};
      HIP_TRY(hipMemcpyAsync(wanted, misc, 16, hipMemcpyDeviceToHost, s));
      HIP_TRY(hipStreamSynchronize(s));
      if (wanted[0] <= capacity && wanted[1] <= capacity_long) {
        total_records += wanted[0] + wanted[1];
        break;
      }
      if (attempt) return fail(TCRC_ERR_HIP, "log scan: descriptor count changed between passes");
      capacity = wanted[0] > capacity ? wanted[0] : capacity;
      capacity_long = wanted[1] > capacity_long ? wanted[1] : capacity_long;
    }
  }
  if (h_records) *h_records = total_records;
  return TCRC_OK;
}
