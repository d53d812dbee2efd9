// Host-pointer entry points: copy in, run the device path, copy the (small) results out.
// The hot path of the reference starts and ends in host memory (log buffers, blocks read
// through env/io_posix.cc), so these are what a host caller uses; their rate is bounded
// by the host link, not by the kernels.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <string>
#include <thread>
#include <vector>

#include "tcrc_internal.h"

using namespace tcrc_internal;

namespace {

int host_begin(tcrc_lane* L) {  // the caller has made the context's GPU current
  if (!L->stream) HIP_TRY(hipStreamCreateWithFlags(&L->stream, hipStreamNonBlocking));
  return TCRC_OK;
}

// Leases a lane of the context for one host call: a free one if there is one, a new one while fewer than
// kMaxLanes exist, else the call waits for the lane its thread hashes to.
struct LaneLease {
  tcrc_lane* lane = nullptr;
  explicit LaneLease(tcrc_ctx* ctx) {
    size_t pick = 0;
    {
      std::lock_guard<std::mutex> lock(ctx->lanes_mu);
      for (tcrc_lane* l : ctx->lanes)
        if (l->mu.try_lock()) {
          lane = l;
          return;
        }
      if (ctx->lanes.size() < (size_t)tcrc_ctx::kMaxLanes) {
        lane = new tcrc_lane();
        lane->mu.lock();
        ctx->lanes.push_back(lane);
        return;
      }
      pick = std::hash<std::thread::id>()(std::this_thread::get_id()) % ctx->lanes.size();
      lane = ctx->lanes[pick];
    }
    lane->mu.lock();
  }
  ~LaneLease() { lane->mu.unlock(); }
  LaneLease(const LaneLease&) = delete;
  void operator=(const LaneLease&) = delete;
};

// Layout of the zero-copy block
constexpr size_t kZcDescOff = tcrc_ctx::kZcData;
constexpr size_t kZcSuffixOff = kZcDescOff + tcrc_ctx::kZcMaxItems * sizeof(tcrc_desc);
constexpr size_t kZcOutOff = kZcSuffixOff + tcrc_ctx::kZcMaxItems;
constexpr size_t kZcExpectedOff = kZcOutOff + tcrc_ctx::kZcMaxItems * 4;
constexpr size_t kZcOkOff = kZcExpectedOff + tcrc_ctx::kZcMaxItems * 4;
constexpr size_t kZcFlagOff = (kZcOkOff + tcrc_ctx::kZcMaxItems + 63) & ~(size_t)63;  // completion word of small calls
constexpr size_t kZcBytes = kZcFlagOff + 64;
// Above this many input bytes a copy to HBM pays for itself (the kernel reads host memory at link speed).
constexpr size_t kZcMaxInput = (size_t)1 << 20;

int zc_begin(tcrc_lane* L) {
  if (!L->zc) {
    HIP_TRY(hipHostMalloc((void**)&L->zc, kZcBytes, hipHostMallocMapped | hipHostMallocCoherent));
    memset(L->zc + kZcFlagOff, 0, 64);
  }
  return TCRC_OK;
}

// A small call's kernel leaves its results in the lane's zero-copy block and then writes the lane's next
// sequence number to the block's last word; the call watches that word instead of synchronising the stream.
int lane_signal(tcrc_ctx* ctx, tcrc_lane* L, DoneSignal* d) {  // after zc_begin
  *d = DoneSignal{};
  if (!ctx->poll_completion) return TCRC_OK;
  if (!L->done_counter) {
    HIP_TRY(hipMalloc((void**)&L->done_counter, 128));
    // (the lane's own stream: never under a caller's graph capture, where a memset node does not replay right)
    HIP_TRY(hipMemsetAsync(L->done_counter, 0, 128, L->stream));
  }
  if (++L->done_seq == 0) ++L->done_seq;
  d->counter = L->done_counter;
  d->flag = (uint32_t*)(L->zc + kZcFlagOff);
  d->value = L->done_seq;
  return TCRC_OK;
}

hipError_t lane_wait(tcrc_lane* L, const DoneSignal& d) {
  return d.flag ? wait_done(d.flag, d.value, L->stream) : hipStreamSynchronize(L->stream);
}

// True when p is host memory the DMA engines can read directly (allocated or registered pinned);
// *dev, if given, receives the address a kernel uses for it.
bool is_pinned(const void* p, const void** dev = nullptr) {
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) {
    (void)hipGetLastError();  // ordinary pageable memory is "invalid value" to the runtime
    return false;
  }
  if (a.type != hipMemoryTypeHost || !a.devicePointer) return false;
  if (dev) *dev = a.devicePointer;
  return true;
}

// Copies pageable host memory to the device through pinned slots: host threads fill slots while the copy engine
// drains them, which roughly doubles the rate of a plain hipMemcpy from pageable memory (the runtime stages
// through one thread).  The slots and the threads that fill them belong to the context and are shared by its
// lanes: an upload takes free slots as it goes (never more than kStageAhead at a time), hands their filling to
// the pool, and issues each slot's copy on its own lane's stream as soon as the slot is full; a slot goes back to
// the free list when its copy has been issued, with the event that says when the copy is over - whoever takes
// it next waits for that.  Uploads of several threads so share the pool slot by slot instead of taking turns
// at whole copies.
}  // namespace

struct tcrc_stage_pool {
  static constexpr int kSlots = tcrc_ctx::kStageSlots;
  static constexpr size_t kSlotBytes = tcrc_ctx::kStageSlotBytes;
  static constexpr int kWorkers = 8;
  struct Task {
    char* dst;
    const char* src;
    size_t len;
    std::atomic<int>* filled;
  };
  int device = 0;
  char* ring = nullptr;
  hipEvent_t done[kSlots] = {};
  bool recorded[kSlots] = {};  // done[i] has been recorded at least once (read by whoever holds slot i)
  std::mutex mu;
  std::condition_variable slot_free, task_ready;
  std::vector<int> free_slots;
  std::deque<Task> tasks;
  std::vector<std::thread> workers;
  bool stop = false;

  void work() {
    for (;;) {
      Task t;
      {
        std::unique_lock<std::mutex> lock(mu);
        task_ready.wait(lock, [&] { return stop || !tasks.empty(); });
        if (tasks.empty()) return;  // stop
        t = tasks.front();
        tasks.pop_front();
      }
      memcpy(t.dst, t.src, t.len);
      t.filled->store(1, std::memory_order_release);
    }
  }
  // a free slot; with wait == false, -1 when there is none right now
  int take_slot(bool wait) {
    std::unique_lock<std::mutex> lock(mu);
    if (!wait && free_slots.empty()) return -1;
    slot_free.wait(lock, [&] { return !free_slots.empty(); });
    const int s = free_slots.back();
    free_slots.pop_back();
    return s;
  }
  void give_slot(int s) {
    {
      std::lock_guard<std::mutex> lock(mu);
      free_slots.push_back(s);
    }
    slot_free.notify_one();
  }
  void submit(const Task& t) {
    {
      std::lock_guard<std::mutex> lock(mu);
      tasks.push_back(t);
    }
    task_ready.notify_one();
  }
};

namespace tcrc_internal {
void stage_pool_destroy(tcrc_ctx* ctx) {
  tcrc_stage_pool* p = ctx->stage_pool;
  if (!p) return;
  {
    std::lock_guard<std::mutex> lock(p->mu);
    p->stop = true;
  }
  p->task_ready.notify_all();
  for (auto& t : p->workers) t.join();
  for (hipEvent_t ev : p->done)
    if (ev) (void)hipEventDestroy(ev);
  if (p->ring) (void)hipHostFree(p->ring);
  delete p;
  ctx->stage_pool = nullptr;
}
}  // namespace tcrc_internal

namespace {

constexpr int kStageAhead = 6;  // slots one upload has in hand at most (being filled, or full and not yet issued)

int stage_pool_get(tcrc_ctx* ctx, tcrc_stage_pool** out) {
  std::lock_guard<std::mutex> lock(ctx->stage_mu);
  if (!ctx->stage_pool) {
    tcrc_stage_pool* p = new tcrc_stage_pool();
    p->device = ctx->device;
    hipError_t e = hipHostMalloc((void**)&p->ring, tcrc_stage_pool::kSlots * tcrc_stage_pool::kSlotBytes, hipHostMallocDefault);
    for (int i = 0; i < tcrc_stage_pool::kSlots && e == hipSuccess; ++i)
      e = hipEventCreateWithFlags(&p->done[i], hipEventDisableTiming);
    if (e != hipSuccess) {
      for (hipEvent_t ev : p->done)
        if (ev) (void)hipEventDestroy(ev);
      if (p->ring) (void)hipHostFree(p->ring);
      delete p;
      return fail(TCRC_ERR_HIP, std::string("staging slots: ") + hipGetErrorString(e));
    }
    for (int i = 0; i < tcrc_stage_pool::kSlots; ++i) p->free_slots.push_back(i);
    for (int i = 0; i < tcrc_stage_pool::kWorkers; ++i) p->workers.emplace_back([p] { p->work(); });
    ctx->stage_pool = p;
  }
  *out = ctx->stage_pool;
  return TCRC_OK;
}

int staged_upload(tcrc_ctx* ctx, tcrc_lane* L, void* d_dst, const void* h_src, size_t bytes) {
  tcrc_stage_pool* P = nullptr;
  if (int rc = stage_pool_get(ctx, &P)) return rc;
  constexpr size_t S = tcrc_stage_pool::kSlotBytes;
  const size_t n = (bytes + S - 1) / S;
  struct InHand {
    int slot;
    std::atomic<int> filled{0};
  };
  std::deque<InHand> hand;  // in chunk order (a deque never moves its elements: the workers hold pointers to `filled`)
  hipError_t err = hipSuccess;
  size_t next_fill = 0, next_issue = 0;
  while (next_issue < n) {
    // take slots and hand them to the pool while there is something left to fill and room in hand
    while (next_fill < n && hand.size() < (size_t)kStageAhead && err == hipSuccess) {
      // (an upload only waits for a slot when it has none in hand: with some in hand it goes on to issue those,
      // which is what frees slots - uploads that all held a few and all waited for more would never finish)
      const int slot = P->take_slot(hand.empty());
      if (slot < 0) break;
      // the slot may still be the source of a copy that an earlier upload (of any lane) queued
      if (P->recorded[slot]) err = hipEventSynchronize(P->done[slot]);
      hand.emplace_back();
      hand.back().slot = slot;
      if (err != hipSuccess) {
        hand.back().filled.store(1);  // nothing will fill it
        break;
      }
      const size_t off = next_fill * S, len = bytes - off < S ? bytes - off : S;
      P->submit(tcrc_stage_pool::Task{P->ring + (size_t)slot * S, (const char*)h_src + off, len, &hand.back().filled});
      ++next_fill;
    }
    if (err != hipSuccess) break;
    // the oldest slot in hand: wait for it to be full, issue its copy, give it back
    InHand& h = hand.front();
    while (!h.filled.load(std::memory_order_acquire)) std::this_thread::yield();
    const size_t off = next_issue * S, len = bytes - off < S ? bytes - off : S;
    err = hipMemcpyAsync((char*)d_dst + off, P->ring + (size_t)h.slot * S, len, hipMemcpyHostToDevice, L->stream);
    if (err == hipSuccess) err = hipEventRecord(P->done[h.slot], L->stream);
    if (err == hipSuccess) P->recorded[h.slot] = true;
    P->give_slot(h.slot);
    hand.pop_front();
    ++next_issue;
  }
  // on an error: whatever is still in hand is waited for (a worker may be writing into it) and given back
  for (InHand& h : hand) {
    while (!h.filled.load(std::memory_order_acquire)) std::this_thread::yield();
    P->give_slot(h.slot);
  }
  if (err != hipSuccess) return fail(TCRC_ERR_HIP, std::string("staged upload: ") + hipGetErrorString(err));
  return TCRC_OK;
}

// Every descriptor must lie inside [0, base_bytes) together with the stored value the flags place
// `after` bytes behind it or 6 bytes in front of it, and be short enough for one lane group.
int check_descs(const tcrc_desc* d, uint64_t n, uint64_t base_bytes, uint64_t after, bool before) {
  for (uint64_t i = 0; i < n; ++i) {
    if (d[i].len > (1u << 30)) return fail(TCRC_ERR_TOO_LARGE, "descriptor above 1 GiB: use tcrc_dev_extend_long / tcrc_extend");
    if (d[i].offset > base_bytes || d[i].len + after > base_bytes - d[i].offset)
      return fail(TCRC_ERR_INVALID, after ? "descriptor (or its stored checksum) outside the buffer"
                                          : "descriptor outside the buffer");
    if (before && d[i].offset < 6) return fail(TCRC_ERR_INVALID, "descriptor leaves no room for the stored checksum before it");
  }
  return TCRC_OK;
}

int check_flags_host(uint32_t flags, uint32_t allowed) {
  if (flags & ~allowed) return fail(TCRC_ERR_INVALID, "unknown or misplaced flag bits");
  return TCRC_OK;
}

// Small batches are served without copy commands when the input is short enough to bounce through the
// zero-copy block or already lies in pinned memory.
bool zero_copy_fits(const void* h_base, uint64_t base_bytes, uint64_t n) {
  return n <= tcrc_ctx::kZcMaxItems && base_bytes <= kZcMaxInput && (base_bytes <= tcrc_ctx::kZcData || is_pinned(h_base));
}

// A small batch gives every buffer one wave; a buffer of hundreds of KiB would keep that wave busy for
// hundreds of microseconds.  Such batches (a table's filter or index block among its meta blocks) go
// through the long-buffer path instead, which spreads every buffer over the chip.
constexpr uint32_t kLongRouteLen = 80u << 10;  // measured break-even with one wave per buffer: about 64 KiB
bool wants_long_route(const tcrc_desc* d, uint64_t n, const uint8_t* h_suffix) {
  if (h_suffix || n > tcrc_ctx::kZcMaxItems) return false;
  for (uint64_t i = 0; i < n; ++i)
    if (d[i].len >= kLongRouteLen) return true;
  return false;
}

int upload(tcrc_ctx* ctx, tcrc_lane* L, tcrc_scratch* s, const void* h, size_t bytes, size_t slack);

// Plain CRCs of the descriptors' spans through the long-buffer path, into h_crc.  The caller holds lane L.
int long_route(tcrc_ctx* ctx, tcrc_lane* L, const void* h_base, uint64_t base_bytes, const tcrc_desc* h_desc, uint64_t n,
               uint32_t* h_crc) {
  // (n <= kZcMaxItems here: the results land in the zero-copy block, so the call that returns with them
  // complete is the only synchronisation)
  if (int rc = zc_begin(L)) return rc;
  if (int rc = upload(ctx, L, &L->data, h_base, base_bytes, 64)) return rc;
  if (int rc = tcrc_dev_extend_long(ctx, L->data.p, h_desc, n, 0, (uint32_t*)(L->zc + kZcOutOff),
                                    L->stream)) {
    (void)hipStreamSynchronize(L->stream);  // the upload may still be reading the caller's memory
    return rc;
  }
  memcpy(h_crc, L->zc + kZcOutOff, n * 4);
  return TCRC_OK;
}

int upload(tcrc_ctx* ctx, tcrc_lane* L, tcrc_scratch* s, const void* h, size_t bytes, size_t slack = 64) {
  if (int rc = scratch_reserve(ctx, s, bytes + slack)) return rc;
  ctx->st_host_bytes_in.fetch_add(bytes, std::memory_order_relaxed);
  if (bytes >= ((size_t)16 << 20) && !is_pinned(h)) return staged_upload(ctx, L, s->p, h, bytes);
  if (bytes) HIP_TRY(hipMemcpyAsync(s->p, h, bytes, hipMemcpyHostToDevice, L->stream));
  return TCRC_OK;
}

}  // namespace

extern "C" {

void* tcrc_host_alloc(tcrc_ctx* ctx, size_t bytes) {
  if (!ctx) return nullptr;
  if (bytes == 0) bytes = 1;
  {
    // smallest kept buffer that is large enough and not more than twice the request
    std::lock_guard<std::mutex> lock(ctx->pin_mu);
    size_t best = ctx->pin_free.size();
    for (size_t i = 0; i < ctx->pin_free.size(); ++i) {
      const size_t cap = ctx->pin_free[i].second;
      if (cap >= bytes && cap / 2 <= bytes && (best == ctx->pin_free.size() || cap < ctx->pin_free[best].second))
        best = i;
    }
    if (best != ctx->pin_free.size()) {
      std::pair<void*, size_t> e = ctx->pin_free[best];
      ctx->pin_free.erase(ctx->pin_free.begin() + best);
      ctx->pin_free_bytes -= e.second;
      ctx->pin_live.push_back(e);
      return e.first;
    }
  }
  tcrc_internal::DeviceScope scope(ctx->device);
  if (scope.err != hipSuccess) return nullptr;
  void* p = nullptr;
  if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lock(ctx->pin_mu);
  ctx->pin_live.emplace_back(p, bytes);
  return p;
}

void tcrc_host_free(tcrc_ctx* ctx, void* p) {
  if (!p) return;
  if (!ctx) {
    (void)hipHostFree(p);
    return;
  }
  const size_t kKeepBytes = (size_t)1 << 30, kKeepCount = 16;
  size_t cap = 0;
  {
    std::lock_guard<std::mutex> lock(ctx->pin_mu);
    for (size_t i = 0; i < ctx->pin_live.size(); ++i)
      if (ctx->pin_live[i].first == p) {
        cap = ctx->pin_live[i].second;
        ctx->pin_live.erase(ctx->pin_live.begin() + i);
        break;
      }
    if (cap && ctx->pin_free_bytes + cap <= kKeepBytes && ctx->pin_free.size() < kKeepCount) {
      ctx->pin_free.emplace_back(p, cap);
      ctx->pin_free_bytes += cap;
      return;
    }
  }
  (void)hipHostFree(p);
}

int tcrc_host_batch_stamp(tcrc_ctx* ctx, const void* h_base, uint64_t base_bytes, const tcrc_desc* h_desc, uint64_t n,
                          const uint8_t* h_suffix, uint32_t flags, uint32_t* h_out) {
  if (!ctx) return fail(TCRC_ERR_INVALID, "null context");
  if (n == 0) return TCRC_OK;
  if (!h_base || !h_desc || !h_out) return fail(TCRC_ERR_INVALID, "null argument");
  if (flags & TCRC_F_STORE_AFTER) return fail(TCRC_ERR_INVALID, "host entry points return the values: nothing to store into");
  if (int rc = check_descs(h_desc, n, base_bytes, 0, false)) return rc;
  HostCallTimer timer(ctx);
  LaneLease lease(ctx);
  tcrc_lane* const L = lease.lane;
  TCRC_DEVICE_SCOPE(ctx);
  if (int rc = host_begin(L)) return rc;
  StreamGuard guard(L->stream);  // an error return must not leave copies from the caller's memory in flight
  if (wants_long_route(h_desc, n, h_suffix)) {
    // (the four-lanes hint says nothing on this route: every buffer is cut into segments anyway)
    if (int rc = check_flags_host(flags, TCRC_F_MASK | TCRC_F_SHORT_SPANS)) return rc;
    if (int rc = long_route(ctx, L, h_base, base_bytes, h_desc, n, h_out)) return rc;
    guard.dismiss();  // long_route returns with the kernel's results complete: the copies ahead of it have finished
    if (flags & TCRC_F_MASK)
      for (uint64_t i = 0; i < n; ++i) h_out[i] = tcrc_mask(h_out[i]);
    return TCRC_OK;
  }
  if (zero_copy_fits(h_base, base_bytes, n)) {
    // Small batch: no copies.  The kernel reads the bytes where they are (pinned memory) or from the bounce
    // area, takes its descriptors from the zero-copy block and writes the results there.
    if (int rc = zc_begin(L)) return rc;
    const void* base = nullptr;
    if (!is_pinned(h_base, &base)) {
      memcpy(L->zc, h_base, base_bytes);
      base = L->zc;
    }
    memcpy(L->zc + kZcDescOff, h_desc, n * sizeof(tcrc_desc));
    if (h_suffix) memcpy(L->zc + kZcSuffixOff, h_suffix, n);
    DoneSignal done;
    if (int rc = lane_signal(ctx, L, &done)) return rc;
    if (int rc = batch_stamp(ctx, base, (const tcrc_desc*)(L->zc + kZcDescOff), n,
                             h_suffix ? (const uint8_t*)(L->zc + kZcSuffixOff) : nullptr, flags,
                             (uint32_t*)(L->zc + kZcOutOff), L->stream, done))
      return rc;
    guard.dismiss();
    HIP_TRY(lane_wait(L, done));
    memcpy(h_out, L->zc + kZcOutOff, n * 4);
    return TCRC_OK;
  }
  if (int rc = upload(ctx, L, &L->data, h_base, base_bytes)) return rc;
  if (int rc = upload(ctx, L, &L->desc, h_desc, n * sizeof(tcrc_desc))) return rc;
  if (h_suffix)
    if (int rc = upload(ctx, L, &L->aux, h_suffix, n)) return rc;
  if (int rc = scratch_reserve(ctx, &L->out, n * 4)) return rc;
  if (int rc = tcrc_dev_batch_stamp(ctx, L->data.p, (const tcrc_desc*)L->desc.p, n,
                                    h_suffix ? (const uint8_t*)L->aux.p : nullptr, flags,
                                    (uint32_t*)L->out.p, L->stream))
    return rc;
  HIP_TRY(hipMemcpyAsync(h_out, L->out.p, n * 4, hipMemcpyDeviceToHost, L->stream));
  guard.dismiss();
  HIP_TRY(hipStreamSynchronize(L->stream));
  return TCRC_OK;
}

int tcrc_host_blocks_stamp(tcrc_ctx* ctx, const void* h_base, uint64_t n_blocks, uint32_t block_len, uint64_t stride,
                           uint32_t init, int suffix_byte, uint32_t flags, uint32_t* h_out) {
  if (!ctx) return fail(TCRC_ERR_INVALID, "null context");
  if (n_blocks == 0) return TCRC_OK;
  if (!h_base || !h_out) return fail(TCRC_ERR_INVALID, "null argument");
  if (flags & TCRC_F_STORE_AFTER) return fail(TCRC_ERR_INVALID, "host entry points return the values: nothing to store into");
  // the geometry is checked before anything is read from the caller's memory
  if (block_len > (1u << 30)) return fail(TCRC_ERR_TOO_LARGE, "block_len above 1 GiB: use the long-buffer entry point");
  if (n_blocks > 1 && stride < block_len) return fail(TCRC_ERR_INVALID, "stride smaller than block_len");
  if (n_blocks > 1 && (n_blocks - 1) > (~0ull - block_len) / stride)
    return fail(TCRC_ERR_INVALID, "n_blocks * stride overflows");
  HostCallTimer timer(ctx);
  LaneLease lease(ctx);
  tcrc_lane* const L = lease.lane;
  TCRC_DEVICE_SCOPE(ctx);
  if (int rc = host_begin(L)) return rc;
  StreamGuard guard(L->stream);  // an error return must not leave copies from the caller's memory in flight
  const uint64_t bytes = (n_blocks - 1) * stride + block_len;
  if (int rc = upload(ctx, L, &L->data, h_base, bytes)) return rc;
  if (int rc = scratch_reserve(ctx, &L->out, n_blocks * 4)) return rc;
  if (int rc = tcrc_dev_blocks_stamp(ctx, L->data.p, n_blocks, block_len, stride, init, suffix_byte, nullptr,
                                     flags, (uint32_t*)L->out.p, L->stream))
    return rc;
  HIP_TRY(hipMemcpyAsync(h_out, L->out.p, n_blocks * 4, hipMemcpyDeviceToHost, L->stream));
  guard.dismiss();
  HIP_TRY(hipStreamSynchronize(L->stream));
  return TCRC_OK;
}

int tcrc_host_blocks_verify(tcrc_ctx* ctx, const void* h_base, uint64_t n_blocks, uint32_t block_len, uint64_t stride,
                            uint32_t init, int suffix_byte, uint32_t flags, const uint32_t* h_expected, uint32_t* h_out,
                            uint8_t* h_ok, uint64_t* h_counts) {
  if (!ctx) return fail(TCRC_ERR_INVALID, "null context");
  if (!h_counts) return fail(TCRC_ERR_INVALID, "verify needs a counts block");
  h_counts[TCRC_CNT_MISMATCH] = 0;
  h_counts[TCRC_CNT_FIRST_BAD] = ~0ull;
  h_counts[TCRC_CNT_CHECKED] = 0;
  if (n_blocks == 0) return TCRC_OK;
  if (!h_base) return fail(TCRC_ERR_INVALID, "null argument");
  // the geometry is checked before anything is read from the caller's memory
  if (block_len > (1u << 30)) return fail(TCRC_ERR_TOO_LARGE, "block_len above 1 GiB: use the long-buffer entry point");
  if (n_blocks > 1 && stride < block_len) return fail(TCRC_ERR_INVALID, "stride smaller than block_len");
  if (n_blocks > 1 && (n_blocks - 1) > (~0ull - block_len - 8) / stride) return fail(TCRC_ERR_INVALID, "n_blocks * stride overflows");
  if (!h_expected && !(flags & (TCRC_F_EXPECT_AFTER | TCRC_F_EXPECT_BEFORE)))
    return fail(TCRC_ERR_INVALID, "verify needs h_expected or an in-image expectation flag");
  if ((flags & TCRC_F_EXPECT_BEFORE)) return fail(TCRC_ERR_INVALID, "uniform blocks keep their stored value behind them");
  HostCallTimer timer(ctx);
  LaneLease lease(ctx);
  tcrc_lane* const L = lease.lane;
  TCRC_DEVICE_SCOPE(ctx);
  if (int rc = host_begin(L)) return rc;
  StreamGuard guard(L->stream);  // an error return must not leave copies from the caller's memory in flight
  // the image's last block is followed by its stored value (and suffix byte) when that is where it is expected
  const uint64_t tail = (flags & TCRC_F_EXPECT_AFTER) ? (suffix_byte >= 0 ? 5u : 4u) : 0u;
  const uint64_t bytes = (n_blocks - 1) * stride + block_len + tail;
  if (int rc = upload(ctx, L, &L->data, h_base, bytes)) return rc;
  // aux layout: [counts 3 x u64 | pad][expected n x u32][ok n]
  const size_t off_expected = 32, off_ok = off_expected + n_blocks * 4;
  if (int rc = scratch_reserve(ctx, &L->aux, off_ok + n_blocks + 64)) return rc;
  uint8_t* aux = (uint8_t*)L->aux.p;
  if (h_expected) HIP_TRY(hipMemcpyAsync(aux + off_expected, h_expected, n_blocks * 4, hipMemcpyHostToDevice, L->stream));
  if (int rc = scratch_reserve(ctx, &L->out, n_blocks * 4)) return rc;
  if (int rc = tcrc_dev_counts_reset(ctx, (uint64_t*)aux, L->stream)) return rc;
  if (int rc = tcrc_dev_blocks_verify(ctx, L->data.p, n_blocks, block_len, stride, init, suffix_byte, nullptr, flags,
                                      h_expected ? (const uint32_t*)(aux + off_expected) : nullptr,
                                      (uint32_t*)L->out.p, aux + off_ok, (uint64_t*)aux, L->stream))
    return rc;
  if (h_out) HIP_TRY(hipMemcpyAsync(h_out, L->out.p, n_blocks * 4, hipMemcpyDeviceToHost, L->stream));
  if (h_ok) HIP_TRY(hipMemcpyAsync(h_ok, aux + off_ok, n_blocks, hipMemcpyDeviceToHost, L->stream));
  HIP_TRY(hipMemcpyAsync(h_counts, aux, 24, hipMemcpyDeviceToHost, L->stream));
  guard.dismiss();
  HIP_TRY(hipStreamSynchronize(L->stream));
  return TCRC_OK;
}

int tcrc_host_batch_verify(tcrc_ctx* ctx, const void* h_base, uint64_t base_bytes, const tcrc_desc* h_desc, uint64_t n,
                           const uint8_t* h_suffix, uint32_t flags, const uint32_t* h_expected, uint32_t* h_out,
                           uint8_t* h_ok, uint64_t* h_counts) {
  if (!ctx) return fail(TCRC_ERR_INVALID, "null context");
  if (!h_counts) return fail(TCRC_ERR_INVALID, "verify needs a counts block");
  h_counts[TCRC_CNT_MISMATCH] = 0;
  h_counts[TCRC_CNT_FIRST_BAD] = ~0ull;
  h_counts[TCRC_CNT_CHECKED] = 0;
  if (n == 0) return TCRC_OK;
  if (!h_base || !h_desc) return fail(TCRC_ERR_INVALID, "null argument");
  if (int rc = check_descs(h_desc, n, base_bytes, (flags & TCRC_F_EXPECT_AFTER) ? (h_suffix ? 5 : 4) : 0,
                           (flags & TCRC_F_EXPECT_BEFORE) != 0))
    return rc;
  HostCallTimer timer(ctx);
  LaneLease lease(ctx);
  tcrc_lane* const L = lease.lane;
  TCRC_DEVICE_SCOPE(ctx);
  if (int rc = host_begin(L)) return rc;
  StreamGuard guard(L->stream);  // an error return must not leave copies from the caller's memory in flight
  if (wants_long_route(h_desc, n, h_suffix)) {
    if (int rc = check_flags_host(flags, TCRC_F_MASK | TCRC_F_EXPECT_AFTER | TCRC_F_EXPECT_BEFORE | TCRC_F_SHORT_SPANS))
      return rc;
    if (!h_expected && !(flags & (TCRC_F_EXPECT_AFTER | TCRC_F_EXPECT_BEFORE)))
      return fail(TCRC_ERR_INVALID, "verify needs h_expected or an in-image expectation flag");
    std::vector<uint32_t> crc(n);
    if (int rc = long_route(ctx, L, h_base, base_bytes, h_desc, n, crc.data())) return rc;
    guard.dismiss();  // long_route returns with the kernel's results complete: the copies ahead of it have finished
    for (uint64_t i = 0; i < n; ++i) {
      const uint32_t masked = tcrc_mask(crc[i]);
      uint32_t stored;
      if (h_expected) {
        stored = h_expected[i];
      } else {
        const uint8_t* e = (const uint8_t*)h_base + h_desc[i].offset;
        e = (flags & TCRC_F_EXPECT_AFTER) ? e + h_desc[i].len : e - 6;
        stored = (uint32_t)e[0] | ((uint32_t)e[1] << 8) | ((uint32_t)e[2] << 16) | ((uint32_t)e[3] << 24);
      }
      const bool good = stored == masked;
      if (h_out) h_out[i] = masked;
      if (h_ok) h_ok[i] = good ? 1 : 0;
      if (!good && h_counts[TCRC_CNT_MISMATCH]++ == 0) h_counts[TCRC_CNT_FIRST_BAD] = i;
    }
    h_counts[TCRC_CNT_CHECKED] = n;
    return TCRC_OK;
  }
  if (zero_copy_fits(h_base, base_bytes, n)) {
    // Small batch (one block fetched by a read, a handful of meta blocks): no copy commands.  The kernel
    // reads the bytes where they lie or from the bounce area and leaves values and verdicts in the
    // zero-copy block; the three counters are derived from the verdicts here, so that the kernel's own
    // counter updates (atomics) stay in device memory.
    if (int rc = zc_begin(L)) return rc;
    if (int rc = scratch_reserve(ctx, &L->aux, 64)) return rc;
    const void* base = nullptr;
    if (!is_pinned(h_base, &base)) {
      memcpy(L->zc, h_base, base_bytes);
      base = L->zc;
    }
    memcpy(L->zc + kZcDescOff, h_desc, n * sizeof(tcrc_desc));
    if (h_suffix) memcpy(L->zc + kZcSuffixOff, h_suffix, n);
    if (h_expected) memcpy(L->zc + kZcExpectedOff, h_expected, n * 4);
    const uint8_t* ok = (const uint8_t*)(L->zc + kZcOkOff);
    DoneSignal done;
    if (int rc = lane_signal(ctx, L, &done)) return rc;
    if (int rc = batch_verify(ctx, base, (const tcrc_desc*)(L->zc + kZcDescOff), n,
                              h_suffix ? (const uint8_t*)(L->zc + kZcSuffixOff) : nullptr, flags,
                              h_expected ? (const uint32_t*)(L->zc + kZcExpectedOff) : nullptr,
                              (uint32_t*)(L->zc + kZcOutOff), (uint8_t*)(L->zc + kZcOkOff), (uint64_t*)L->aux.p,
                              L->stream, done))
      return rc;
    guard.dismiss();
    HIP_TRY(lane_wait(L, done));
    if (h_out) memcpy(h_out, L->zc + kZcOutOff, n * 4);
    if (h_ok) memcpy(h_ok, ok, n);
    for (uint64_t i = 0; i < n; ++i)
      if (!ok[i]) {
        if (h_counts[TCRC_CNT_MISMATCH]++ == 0) h_counts[TCRC_CNT_FIRST_BAD] = i;
      }
    h_counts[TCRC_CNT_CHECKED] = n;
    return TCRC_OK;
  }
  if (int rc = upload(ctx, L, &L->data, h_base, base_bytes)) return rc;
  if (int rc = upload(ctx, L, &L->desc, h_desc, n * sizeof(tcrc_desc))) return rc;
  // host_aux layout: [counts 3 x u64][suffix n][pad][expected n x u32][ok n]
  const size_t off_suffix = 32, off_expected = (off_suffix + n + 15) & ~(size_t)15, off_ok = off_expected + n * 4;
  if (int rc = scratch_reserve(ctx, &L->aux, off_ok + n + 64)) return rc;
  uint8_t* aux = (uint8_t*)L->aux.p;
  if (h_suffix) HIP_TRY(hipMemcpyAsync(aux + off_suffix, h_suffix, n, hipMemcpyHostToDevice, L->stream));
  if (h_expected)
    HIP_TRY(hipMemcpyAsync(aux + off_expected, h_expected, n * 4, hipMemcpyHostToDevice, L->stream));
  if (int rc = scratch_reserve(ctx, &L->out, n * 4)) return rc;
  if (int rc = tcrc_dev_counts_reset(ctx, (uint64_t*)aux, L->stream)) return rc;
  if (int rc = tcrc_dev_batch_verify(ctx, L->data.p, (const tcrc_desc*)L->desc.p, n,
                                     h_suffix ? aux + off_suffix : nullptr, flags,
                                     h_expected ? (const uint32_t*)(aux + off_expected) : nullptr,
                                     (uint32_t*)L->out.p, aux + off_ok, (uint64_t*)aux, L->stream))
    return rc;
  if (h_out) HIP_TRY(hipMemcpyAsync(h_out, L->out.p, n * 4, hipMemcpyDeviceToHost, L->stream));
  if (h_ok) HIP_TRY(hipMemcpyAsync(h_ok, aux + off_ok, n, hipMemcpyDeviceToHost, L->stream));
  HIP_TRY(hipMemcpyAsync(h_counts, aux, 24, hipMemcpyDeviceToHost, L->stream));
  guard.dismiss();
  HIP_TRY(hipStreamSynchronize(L->stream));
  return TCRC_OK;
}

int tcrc_host_wal_verify(tcrc_ctx* ctx, const void* h_img, uint64_t size, uint64_t log_number,
                         tcrc_wal_block* h_blocks, uint64_t* h_counts, uint64_t* h_records) {
  if (h_counts) {
    h_counts[TCRC_CNT_MISMATCH] = 0;
    h_counts[TCRC_CNT_FIRST_BAD] = ~0ull;
    h_counts[TCRC_CNT_CHECKED] = 0;
  }
  return tcrc_host_wal_verify_part(ctx, h_img, size, log_number, 0, 0, h_blocks, h_counts, h_records);
}

int tcrc_host_wal_verify_part(tcrc_ctx* ctx, const void* h_part, uint64_t size, uint64_t log_number,
                              uint64_t first_block, int more_follows, tcrc_wal_block* h_blocks, uint64_t* h_counts,
                              uint64_t* h_records) {
  if (!ctx) return fail(TCRC_ERR_INVALID, "null context");
  if (h_records) *h_records = 0;
  if (more_follows && size % 32768 != 0)
    return fail(TCRC_ERR_INVALID, "a part that is not the end of the log must be whole 32 KiB blocks");
  if (size == 0) return TCRC_OK;
  if (!h_part || !h_blocks) return fail(TCRC_ERR_INVALID, "null argument");
  HostCallTimer timer(ctx);
  LaneLease lease(ctx);
  tcrc_lane* const L = lease.lane;
  TCRC_DEVICE_SCOPE(ctx);
  if (int rc = host_begin(L)) return rc;
  StreamGuard guard(L->stream);  // an error return must not leave copies from the caller's memory in flight
  const uint64_t nblocks = (size + 32767) / 32768;
  if (int rc = upload(ctx, L, &L->data, h_part, size)) return rc;
  if (int rc = scratch_reserve(ctx, &L->blocks, nblocks * sizeof(tcrc_wal_block))) return rc;
  if (int rc = scratch_reserve(ctx, &L->aux, 64)) return rc;
  if (int rc = tcrc_dev_counts_reset(ctx, (uint64_t*)L->aux.p, L->stream)) return rc;
  if (int rc = tcrc_dev_wal_verify_part(ctx, L->data.p, size, log_number, first_block, more_follows,
                                        (tcrc_wal_block*)L->blocks.p, (uint64_t*)L->aux.p, h_records,
                                        L->stream))
    return rc;
  HIP_TRY(hipMemcpyAsync(h_blocks, L->blocks.p, nblocks * sizeof(tcrc_wal_block), hipMemcpyDeviceToHost,
                         L->stream));
  uint64_t part_counts[TCRC_CNT_WORDS] = {0, ~0ull, 0};
  if (h_counts) HIP_TRY(hipMemcpyAsync(part_counts, L->aux.p, 24, hipMemcpyDeviceToHost, L->stream));
  guard.dismiss();
  HIP_TRY(hipStreamSynchronize(L->stream));
  if (h_counts) {
    h_counts[TCRC_CNT_MISMATCH] += part_counts[TCRC_CNT_MISMATCH];
    if (part_counts[TCRC_CNT_FIRST_BAD] < h_counts[TCRC_CNT_FIRST_BAD])
      h_counts[TCRC_CNT_FIRST_BAD] = part_counts[TCRC_CNT_FIRST_BAD];
    h_counts[TCRC_CNT_CHECKED] += part_counts[TCRC_CNT_CHECKED];
  }
  return TCRC_OK;
}

int tcrc_host_wal_list(tcrc_ctx* ctx, const void* h_img, uint64_t size, uint64_t log_number, tcrc_wal_block* h_blocks,
                       tcrc_wal_record* h_records, uint64_t capacity, uint64_t* h_count) {
  if (!ctx) return fail(TCRC_ERR_INVALID, "null context");
  if (h_count) *h_count = 0;
  if (size == 0) return TCRC_OK;
  if (!h_img || (capacity && !h_records)) return fail(TCRC_ERR_INVALID, "null argument");
  HostCallTimer timer(ctx);
  LaneLease lease(ctx);
  tcrc_lane* const L = lease.lane;
  TCRC_DEVICE_SCOPE(ctx);
  if (int rc = host_begin(L)) return rc;
  StreamGuard guard(L->stream);  // an error return must not leave copies from the caller's memory in flight
  const uint64_t nblocks = (size + 32767) / 32768;
  if (int rc = upload(ctx, L, &L->data, h_img, size)) return rc;
  if (int rc = scratch_reserve(ctx, &L->blocks, nblocks * sizeof(tcrc_wal_block))) return rc;
  if (int rc = scratch_reserve(ctx, &L->aux, 64)) return rc;
  if (int rc = scratch_reserve(ctx, &L->records, (capacity ? capacity : 1) * sizeof(tcrc_wal_record))) return rc;
  if (int rc = tcrc_dev_counts_reset(ctx, (uint64_t*)L->aux.p, L->stream)) return rc;
  if (int rc = tcrc_dev_wal_verify(ctx, L->data.p, size, log_number, (tcrc_wal_block*)L->blocks.p,
                                   (uint64_t*)L->aux.p, nullptr, L->stream))
    return rc;
  uint64_t total = 0;
  if (int rc = tcrc_dev_wal_list(ctx, L->data.p, size, log_number, (const tcrc_wal_block*)L->blocks.p,
                                 (tcrc_wal_record*)L->records.p, capacity, &total, L->stream))
    return rc;
  if (h_blocks)
    HIP_TRY(hipMemcpyAsync(h_blocks, L->blocks.p, nblocks * sizeof(tcrc_wal_block), hipMemcpyDeviceToHost,
                           L->stream));
  const uint64_t n_out = total < capacity ? total : capacity;
  if (n_out)
    HIP_TRY(hipMemcpyAsync(h_records, L->records.p, n_out * sizeof(tcrc_wal_record), hipMemcpyDeviceToHost,
                           L->stream));
  guard.dismiss();
  HIP_TRY(hipStreamSynchronize(L->stream));
  if (h_count) *h_count = total;
  return TCRC_OK;
}

int tcrc_extend(tcrc_ctx* ctx, uint32_t init, const void* h_data, size_t n, uint32_t* out) {
  if (!ctx) return fail(TCRC_ERR_INVALID, "null context");
  if (!out) return fail(TCRC_ERR_INVALID, "null argument");
  *out = init;  // Extend over no bytes leaves the value as it is
  if (n == 0) return TCRC_OK;
  if (!h_data) return fail(TCRC_ERR_INVALID, "null argument");
  HostCallTimer timer(ctx);
  LaneLease lease(ctx);
  tcrc_lane* const L = lease.lane;
  TCRC_DEVICE_SCOPE(ctx);
  if (int rc = host_begin(L)) return rc;
  StreamGuard guard(L->stream);  // an error return must not leave copies from the caller's memory in flight
  // Pieces of up to 256 MiB, chained through init like the callers that feed a file through
  // Extend buffer by buffer (utilities/backupable/backupable_db.cc:1434-1449).
  const size_t piece = (size_t)256 << 20;
  if (int rc = zc_begin(L)) return rc;
  uint32_t crc = init;
  if (n <= (16u << 10)) {
    // short input: one wave walks it in a single launch, reading it from the zero-copy block (or where it
    // lies, if that is pinned memory) and leaving the result there - no copy commands at all
    const void* base = nullptr;
    if (!is_pinned(h_data, &base)) {
      memcpy(L->zc, h_data, n);
      base = L->zc;
    }
    ctx->st_host_bytes_in.fetch_add(n, std::memory_order_relaxed);
    tcrc_desc* const d = (tcrc_desc*)(L->zc + kZcDescOff);
    d->offset = 0;
    d->len = (uint32_t)n;
    d->init = init;
    DoneSignal done;
    if (int rc = lane_signal(ctx, L, &done)) return rc;
    if (int rc = batch_stamp(ctx, base, d, 1, nullptr, 0, (uint32_t*)(L->zc + kZcOutOff), L->stream, done)) return rc;
    guard.dismiss();
    HIP_TRY(lane_wait(L, done));
    memcpy(&crc, L->zc + kZcOutOff, 4);
    *out = crc;
    return TCRC_OK;
  }
  for (size_t done = 0; done < n;) {
    const size_t len = n - done < piece ? n - done : piece;
    if (int rc = upload(ctx, L, &L->data, (const uint8_t*)h_data + done, len)) return rc;
    tcrc_desc d;
    d.offset = 0;
    d.len = (uint32_t)len;
    d.init = crc;
    // the result lands in the zero-copy block: the call returns with it complete, no copy back
    if (int rc = tcrc_dev_extend_long(ctx, L->data.p, &d, 1, 0, (uint32_t*)(L->zc + kZcOutOff),
                                      L->stream)) {
      (void)hipStreamSynchronize(L->stream);  // the upload may still be reading the caller's memory
      return rc;
    }
    memcpy(&crc, L->zc + kZcOutOff, 4);
    done += len;
  }
  guard.dismiss();  // every piece's call returned with its result complete: the copy ahead of it has finished
  *out = crc;
  return TCRC_OK;
}

int tcrc_value(tcrc_ctx* ctx, const void* h_data, size_t n, uint32_t* out) {
  return tcrc_extend(ctx, 0, h_data, n, out);
}

// ---- the reference's scalar signature (util/crc32c.cc:439,1225-1228) over a process-wide context ----
namespace {
std::mutex g_scalar_mu;
tcrc_ctx* g_scalar_ctx = nullptr;
bool g_scalar_tried = false;
std::string g_scalar_init_error;
std::atomic<tcrc_extend_fn> g_scalar_fallback{nullptr};
std::atomic<uint64_t> g_scalar_fallback_calls{0};

tcrc_ctx* scalar_context() {
  std::lock_guard<std::mutex> lock(g_scalar_mu);
  if (!g_scalar_tried) {
    g_scalar_tried = true;
    int device = 0;
    if (const char* e = getenv("TCRC_DEVICE")) device = atoi(e);
    if (tcrc_init(device, &g_scalar_ctx) != TCRC_OK) {
      g_scalar_ctx = nullptr;
      g_scalar_init_error = tcrc_last_error();
    }
  }
  return g_scalar_ctx;
}
}  // namespace

void tcrc_set_scalar_fallback(tcrc_extend_fn fn) { g_scalar_fallback.store(fn); }

uint64_t tcrc_scalar_fallback_calls(void) { return g_scalar_fallback_calls.load(); }

uint32_t tcrc_crc32c_extend(uint32_t init, const char* data, size_t n) {
  if (n == 0) return init;  // util/crc32c.cc:356-398 over no bytes
  std::string why;
  if (tcrc_ctx* ctx = scalar_context()) {
    uint32_t out = 0;
    if (tcrc_extend(ctx, init, data, n, &out) == TCRC_OK) return out;
    why = tcrc_last_error();
  } else {
    why = g_scalar_init_error;
  }
  if (tcrc_extend_fn fb = g_scalar_fallback.load()) {
    g_scalar_fallback_calls.fetch_add(1, std::memory_order_relaxed);
    return fb(init, data, n);
  }
  fprintf(stderr, "tcrc_crc32c_extend: %s, and no fallback was set with tcrc_set_scalar_fallback\n", why.c_str());
  abort();
}

}  // extern "C"
