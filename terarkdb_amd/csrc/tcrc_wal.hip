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

#include <atomic>
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

// The fragmentation rules of db/log_writer.cc:45-100 over a list of record lengths: calls
// f(source offset, image offset of the header, payload bytes, type, zero bytes in front of the header) for every
// physical record in file order and returns the totals.
struct FrameTotals {
  uint64_t image_bytes = 0, physical = 0, entries = 0;
  uint32_t block_offset = 0;
};
template <typename F>
FrameTotals for_each_fragment(const tcrc_desc* recs, uint64_t n, bool recycle, uint32_t block_offset, F&& f) {
  const uint32_t hs = recycle ? kRecyclableHeader : kHeader;
  const uint32_t base_type = recycle ? 4u : 0u;  // recyclable types are the legacy ones + 4 (db/log_format.h:19-35)
  uint64_t pos = 0, phys = 0;
  uint32_t bo = block_offset;
  for (uint64_t i = 0; i < n; ++i) {
    uint64_t src = recs[i].offset;
    uint32_t left = recs[i].len;
    bool first = true;
    for (;;) {
      uint32_t pad = 0;
      const uint32_t room = kLogBlock - bo;
      if (room < hs) {  // a block tail too short for a header is filled with zeros (db/log_writer.cc:61-73)
        pad = room;
        pos += room;
        bo = 0;
      }
      const uint32_t avail = kLogBlock - bo - hs;
      const uint32_t take = left < avail ? left : avail;
      const bool last = take == left;
      const uint32_t t = (first ? (last ? 1u : 2u) : (last ? 4u : 3u)) + base_type;  // full, first, last, middle
      f(src, pos, take, t, pad);
      pos += hs + take;
      bo += hs + take;
      src += take;
      left -= take;
      first = false;
      ++phys;
      if (left == 0) break;
    }
  }
  FrameTotals t;
  t.image_bytes = pos;
  t.physical = phys;
  // an image that ends exactly at a block boundary continues at offset 0 of the next block (the writer's own
  // counter stands at 32768 then, db/log_writer.cc:58-80 turns it into 0 at the next record)
  t.block_offset = bo == kLogBlock ? 0u : bo;
  return t;
}

// The kernel's piece list for these records.  A fragment is cut, from its end, into pieces of kFramePiece bytes
// (the first one may be shorter): at most eight, which all go into ONE round of eight entries, so that the wave
// that takes the round has the whole fragment.  Entries need not follow the file's order - each says where its
// bytes come from and go to - so the list is laid out for the kernel: first the rounds that hold fragments of
// several pieces (packed in file order; a fragment that does not fit the open round closes it), then the
// one-piece fragments, longest first in classes of 256 bytes, filling the holes of those rounds and then rounds
// of their own - a round takes as long as its longest entry, and entries of a class are alike.
constexpr uint32_t kFrameClasses = kFramePiece / 256u + 1u;
struct FrameLayout {
  uint64_t multi_rounds = 0, singles = 0, holes = 0;
  uint64_t class_first[kFrameClasses + 1] = {};  // position of a class's first fragment in the order of one-piece fragments
  std::vector<uint32_t> hole_slots;              // entry indices of the holes in the rounds of several pieces, in order
};
inline uint32_t frame_pieces(uint32_t take) { return take == 0 ? 1u : (take + kFramePiece - 1u) / kFramePiece; }
inline uint32_t frame_class(uint32_t take) { return kFrameClasses - 1u - (take + 255u) / 256u; }  // longest first

FrameTotals frame_layout(const tcrc_desc* recs, uint64_t n, bool recycle, uint32_t block_offset, FrameLayout* lay) {
  uint64_t per_class[kFrameClasses] = {};
  uint32_t free_slots = 0;  // of the open round of several pieces
  FrameTotals t = for_each_fragment(recs, n, recycle, block_offset,
                                    [&](uint64_t, uint64_t, uint32_t take, uint32_t, uint32_t) {
    const uint32_t c = frame_pieces(take);
    if (c == 1u) {
      ++per_class[frame_class(take)];
      return;
    }
    if (c > free_slots) {  // the open round's holes are left for one-piece fragments
      for (uint32_t k = 0; k < free_slots; ++k) lay->hole_slots.push_back((uint32_t)(lay->multi_rounds * 8u - free_slots + k));
      ++lay->multi_rounds;
      free_slots = 8u;
    }
    free_slots -= c;
  });
  for (uint32_t k = 0; k < free_slots; ++k) lay->hole_slots.push_back((uint32_t)(lay->multi_rounds * 8u - free_slots + k));
  lay->holes = lay->hole_slots.size();
  uint64_t at = 0;
  for (uint32_t cl = 0; cl < kFrameClasses; ++cl) {
    lay->class_first[cl] = at;
    at += per_class[cl];
  }
  lay->class_first[kFrameClasses] = at;
  lay->singles = at;
  const uint64_t beyond = at > lay->holes ? at - lay->holes : 0;
  t.entries = lay->multi_rounds * 8u + (beyond + 7u) / 8u * 8u;
  return t;
}

void frame_emit(const tcrc_desc* recs, uint64_t n, bool recycle, uint32_t block_offset, const FrameLayout& lay,
                uint64_t entries, WalFrameEntry* out) {
  const uint32_t hs = recycle ? kRecyclableHeader : kHeader;
  // what no fragment will fill: holes beyond the one-piece fragments, and the end of the last round
  for (uint64_t k = lay.singles; k < lay.holes; ++k) out[lay.hole_slots[k]] = WalFrameEntry{0, 0};
  const uint64_t beyond = lay.singles > lay.holes ? lay.singles - lay.holes : 0;
  for (uint64_t k = lay.multi_rounds * 8u + beyond; k < entries; ++k) out[k] = WalFrameEntry{0, 0};
  uint64_t class_next[kFrameClasses];
  for (uint32_t cl = 0; cl < kFrameClasses; ++cl) class_next[cl] = lay.class_first[cl];
  uint64_t multi_round = 0;
  uint32_t free_slots = 0;
  bool any_multi = false;
  for_each_fragment(recs, n, recycle, block_offset,
                    [&](uint64_t src, uint64_t pos, uint32_t take, uint32_t type, uint32_t pad) {
    const uint32_t c = frame_pieces(take);
    uint64_t slot;
    if (c == 1u) {
      const uint64_t k = class_next[frame_class(take)]++;
      slot = k < lay.holes ? lay.hole_slots[k] : lay.multi_rounds * 8u + (k - lay.holes);
    } else {
      if (c > free_slots) {
        if (any_multi) ++multi_round;
        any_multi = true;
        free_slots = 8u;
      }
      slot = multi_round * 8u + (8u - free_slots);
      free_slots -= c;
    }
    uint64_t s_off = src, d_off = pos + hs;
    for (uint32_t k = 0; k < c; ++k) {
      const uint32_t q = c - 1u - k;
      const uint32_t plen = k == 0 ? take - q * kFramePiece : kFramePiece;
      out[slot + k].a = s_off | ((uint64_t)plen << 40) | ((uint64_t)q << 53) | ((uint64_t)(k == 0) << 56) |
                        ((uint64_t)(q == 0) << 57) | (1ull << 58) | ((uint64_t)(k == 0 ? pad : 0u) << 59);
      out[slot + k].b = d_off | ((uint64_t)take << 40) | ((uint64_t)type << 56);
      s_off += plen;
      d_off += plen;
    }
  });
}

}  // namespace

// A framing plan: where every header and every payload byte of a batch of records goes (host arithmetic on the
// record lengths), with its piece list on its way to, or already in, device memory.
struct tcrc_wal_plan {
  tcrc_ctx* ctx = nullptr;
  tcrc_ctx::FrameBuf buf;        // pinned and device copy of the piece list
  const WalFrameEntry* d_entries = nullptr;  // what the kernel reads (the pinned copy itself for a short list)
  hipEvent_t uploaded = nullptr;  // the copy to the device has finished (NULL: nothing was copied)
  hipEvent_t used = nullptr;      // the last launch that reads the list has finished
  bool used_recorded = false;
  std::atomic<bool> upload_seen{false};  // the host has seen the upload event complete
  std::atomic<bool> caller_waits{false};  // tcrc_wal_plan_caller_waits: no event behind the launches
  FrameTotals totals;
  uint64_t src_end = 0;           // one past the last source byte any record names
  uint32_t header_size = 0, log_number = 0;
};

namespace {
constexpr size_t kFrameZeroCopyEntries = 4096;  // up to 64 KiB of entries are read where they lie, in pinned memory
}

extern "C" uint64_t tcrc_wal_framed_size(const tcrc_desc* h_recs, uint64_t n, int recycle, uint32_t block_offset) {
  if (!h_recs || block_offset >= kLogBlock) return 0;
  return for_each_fragment(h_recs, n, recycle != 0, block_offset, [](uint64_t, uint64_t, uint32_t, uint32_t, uint32_t) {})
      .image_bytes;
}

// Development aid (not part of the C ABI in include/; needs no GPU): the piece list a plan for these records would
// hold, as pairs of 64-bit words, and the number of entries per run; *count receives the number of entries.
extern "C" int tcrc_debug_wal_plan_entries(const tcrc_desc* h_recs, uint64_t n, int recycle, uint32_t block_offset,
                                           uint64_t* out, uint64_t capacity, uint64_t* count, uint32_t* run_entries) {
  if (!count || (n && !h_recs) || block_offset >= kLogBlock) return fail(TCRC_ERR_INVALID, "bad argument");
  FrameLayout lay;
  const FrameTotals t = frame_layout(h_recs, n, recycle != 0, block_offset, &lay);
  *count = t.entries;
  if (run_entries) *run_entries = 8;
  if (out) {
    if (capacity < t.entries) return fail(TCRC_ERR_INVALID, "capacity too small");
    frame_emit(h_recs, n, recycle != 0, block_offset, lay, t.entries, (WalFrameEntry*)out);
  }
  return TCRC_OK;
}

extern "C" void tcrc_wal_plan_free(tcrc_wal_plan* plan) {
  if (!plan) return;
  tcrc_ctx* ctx = plan->ctx;
  tcrc_internal::DeviceScope scope(ctx->device);
  // the buffers go back to the context only when nothing reads them any more
  if (plan->uploaded) (void)hipEventSynchronize(plan->uploaded);
  if (plan->used_recorded) (void)hipEventSynchronize(plan->used);
  if (plan->uploaded) (void)hipEventDestroy(plan->uploaded);
  if (plan->used) (void)hipEventDestroy(plan->used);
  if (plan->buf.h) {
    std::lock_guard<std::mutex> lock(ctx->frame_mu);
    if (ctx->frame_free.size() < 8) {
      ctx->frame_free.push_back(plan->buf);
    } else {
      (void)hipHostFree(plan->buf.h);
      if (plan->buf.d) (void)hipFree(plan->buf.d);
    }
  }
  delete plan;
}

extern "C" int tcrc_wal_plan_create(tcrc_ctx* ctx, const tcrc_desc* h_recs, uint64_t n, uint64_t log_number, int recycle,
                                    uint32_t block_offset, tcrc_wal_plan** out) {
  if (!ctx || !out) return fail(TCRC_ERR_INVALID, "null argument");
  *out = nullptr;
  if (block_offset >= kLogBlock) return fail(TCRC_ERR_INVALID, "block_offset must be below 32768");
  if (n && !h_recs) return fail(TCRC_ERR_INVALID, "null argument");
  // first pass: sizes (and the limits of the entry format)
  uint64_t src_end = 0;
  for (uint64_t i = 0; i < n; ++i) {
    const uint64_t e = h_recs[i].offset + h_recs[i].len;
    if (e < h_recs[i].offset || e > kFrameMaxOffset) return fail(TCRC_ERR_TOO_LARGE, "record beyond 1 TiB of the source");
    if (e > src_end) src_end = e;
  }
  FrameLayout lay;
  const FrameTotals t = frame_layout(h_recs, n, recycle != 0, block_offset, &lay);
  if (t.entries >= (1ull << 32)) return fail(TCRC_ERR_TOO_LARGE, "too many physical records in one plan");
  if (t.image_bytes > kFrameMaxOffset) return fail(TCRC_ERR_TOO_LARGE, "framed image above 1 TiB");
  tcrc_wal_plan* plan = new tcrc_wal_plan();
  plan->ctx = ctx;
  plan->totals = t;
  plan->src_end = src_end;
  plan->header_size = recycle ? kRecyclableHeader : kHeader;
  plan->log_number = (uint32_t)log_number;
  if (t.entries == 0) {
    *out = plan;
    return TCRC_OK;
  }
  tcrc_internal::DeviceScope scope(ctx->device);
  auto bail = [&](int code, const std::string& msg) {
    tcrc_wal_plan_free(plan);
    return fail(code, msg);
  };
  if (scope.err != hipSuccess) return bail(TCRC_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(scope.err));
  const size_t bytes = (size_t)t.entries * sizeof(WalFrameEntry);
  const bool zero_copy = t.entries <= kFrameZeroCopyEntries;
  {
    // smallest kept buffer that is large enough (and has a device side if one is needed)
    std::lock_guard<std::mutex> lock(ctx->frame_mu);
    size_t best = ctx->frame_free.size();
    for (size_t i = 0; i < ctx->frame_free.size(); ++i) {
      const auto& fb = ctx->frame_free[i];
      if (fb.bytes >= bytes && (zero_copy || fb.d) && (best == ctx->frame_free.size() || fb.bytes < ctx->frame_free[best].bytes))
        best = i;
    }
    if (best != ctx->frame_free.size()) {
      plan->buf = ctx->frame_free[best];
      ctx->frame_free.erase(ctx->frame_free.begin() + best);
    }
    if (!ctx->frame_stream) {
      const hipError_t e = hipStreamCreateWithFlags(&ctx->frame_stream, hipStreamNonBlocking);
      if (e != hipSuccess) return bail(TCRC_ERR_HIP, std::string("hipStreamCreate: ") + hipGetErrorString(e));
    }
  }
  if (!plan->buf.h) {
    const size_t want = zero_copy ? kFrameZeroCopyEntries * sizeof(WalFrameEntry) : bytes + bytes / 4;
    hipError_t e = hipHostMalloc(&plan->buf.h, want, hipHostMallocMapped);
    if (e == hipSuccess && !zero_copy) e = hipMalloc(&plan->buf.d, want);
    if (e != hipSuccess) return bail(TCRC_ERR_HIP, std::string("plan buffers: ") + hipGetErrorString(e));
    plan->buf.bytes = want;
  }
  frame_emit(h_recs, n, recycle != 0, block_offset, lay, t.entries, (WalFrameEntry*)plan->buf.h);
  hipError_t e = hipEventCreateWithFlags(&plan->used, hipEventDisableTiming);
  if (e != hipSuccess) return bail(TCRC_ERR_HIP, std::string("hipEventCreate: ") + hipGetErrorString(e));
  if (zero_copy) {
    void* dev = nullptr;
    e = hipHostGetDevicePointer(&dev, plan->buf.h, 0);
    if (e != hipSuccess) return bail(TCRC_ERR_HIP, std::string("hipHostGetDevicePointer: ") + hipGetErrorString(e));
    plan->d_entries = (const WalFrameEntry*)dev;
  } else {
    e = hipEventCreateWithFlags(&plan->uploaded, hipEventDisableTiming);
    if (e == hipSuccess) e = hipMemcpyAsync(plan->buf.d, plan->buf.h, bytes, hipMemcpyHostToDevice, ctx->frame_stream);
    if (e == hipSuccess) e = hipEventRecord(plan->uploaded, ctx->frame_stream);
    if (e != hipSuccess) return bail(TCRC_ERR_HIP, std::string("plan upload: ") + hipGetErrorString(e));
    plan->d_entries = (const WalFrameEntry*)plan->buf.d;
  }
  *out = plan;
  return TCRC_OK;
}

extern "C" int tcrc_wal_plan_caller_waits(tcrc_wal_plan* plan, int yes) {
  if (!plan) return fail(TCRC_ERR_INVALID, "null argument");
  plan->caller_waits.store(yes != 0, std::memory_order_relaxed);
  return TCRC_OK;
}

extern "C" uint64_t tcrc_wal_plan_image_bytes(const tcrc_wal_plan* plan) { return plan ? plan->totals.image_bytes : 0; }
extern "C" uint32_t tcrc_wal_plan_block_offset(const tcrc_wal_plan* plan) { return plan ? plan->totals.block_offset : 0; }
extern "C" uint64_t tcrc_wal_plan_physical_records(const tcrc_wal_plan* plan) { return plan ? plan->totals.physical : 0; }

extern "C" int tcrc_dev_wal_frame_enqueue(tcrc_ctx* ctx, tcrc_wal_plan* plan, const void* d_src, void* d_dst,
                                          uint64_t dst_capacity, void* stream) {
  if (!ctx || !plan) return fail(TCRC_ERR_INVALID, "null argument");
  if (plan->ctx != ctx) return fail(TCRC_ERR_INVALID, "the plan belongs to another context");
  if (plan->totals.entries == 0) return TCRC_OK;
  if (!d_src || !d_dst) return fail(TCRC_ERR_INVALID, "null argument");
  if (plan->totals.image_bytes > dst_capacity) return fail(TCRC_ERR_INVALID, "dst_capacity is smaller than the framed size");
  // the image must not overlap the records it is made from (payload bytes are stored while others are still unread)
  const uintptr_t s0 = (uintptr_t)d_src, s1 = s0 + plan->src_end, t0 = (uintptr_t)d_dst, t1 = t0 + plan->totals.image_bytes;
  if (s0 < t1 && t0 < s1) return fail(TCRC_ERR_INVALID, "d_dst overlaps d_src");
  TCRC_DEVICE_SCOPE(ctx);
  hipStream_t s = (hipStream_t)stream;
  // the launch waits for the list's upload - until the host has once seen that the upload is over (an event in
  // front of every queued launch costs it a couple of microseconds)
  if (plan->uploaded && !plan->upload_seen.load(std::memory_order_acquire)) {
    if (hipEventQuery(plan->uploaded) == hipSuccess) plan->upload_seen.store(true, std::memory_order_release);
    else HIP_TRY(hipStreamWaitEvent(s, plan->uploaded, 0));
  }
  WalFrameParams f{};
  f.src = (const uint8_t*)d_src;
  f.dst = (uint8_t*)d_dst;
  f.entries = plan->d_entries;
  f.nrounds = plan->totals.entries / 8u;
  f.header_size = plan->header_size;
  f.log_number = plan->log_number;
  if (int rc = run_wal_frame(ctx, f, ctx->frame_phase_dst != 0, stream)) return rc;
  if (!plan->caller_waits.load(std::memory_order_relaxed)) {
    // so that the plan can be freed at any time (about 3.5 us per queued launch: tcrc_wal_plan_caller_waits)
    HIP_TRY(hipEventRecord(plan->used, s));
    plan->used_recorded = true;
  }
  return TCRC_OK;
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
  tcrc_wal_plan* plan = nullptr;
  if (int rc = tcrc_wal_plan_create(ctx, h_recs, n, log_number, recycle, block_offset, &plan)) return rc;
  int rc = tcrc_dev_wal_frame_enqueue(ctx, plan, d_src, d_dst, dst_capacity, stream);
  if (rc == TCRC_OK) {
    TCRC_DEVICE_SCOPE(ctx);
    const hipError_t e = hipStreamSynchronize((hipStream_t)stream);  // "returns with the image complete"
    if (e != hipSuccess) rc = fail(TCRC_ERR_HIP, std::string("hipStreamSynchronize: ") + hipGetErrorString(e));
  }
  if (rc == TCRC_OK) {
    if (h_written) *h_written = plan->totals.image_bytes;
    if (block_offset_out) *block_offset_out = plan->totals.block_offset;
    if (h_physical_records) *h_physical_records = plan->totals.physical;
  }
  tcrc_wal_plan_free(plan);
  return rc;
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
