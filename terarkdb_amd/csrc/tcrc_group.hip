// Several GPUs in one process, and the one collective of the process-per-GPU form.
//
// The path shards by units - SST blocks, 32 KiB log blocks, files - that carry their own checksums, so a
// verification corpus splits into contiguous ranges that are verified independently (the loop this replaces is
// DBImpl::VerifyChecksum, db/db_impl.cc:4657-4690, one file after the other on one thread, and inside a file
// BlockBasedTable::VerifyChecksumInBlocks, table/block_based_table_reader.cc:2647-2695, one block after the other).
//   - A tcrc_group owns one context and one stream per member (a member is a GPU; the same GPU may be named
//     more than once) and runs a call's shards on one host thread per member; the three counters are summed /
//     min-ed on the host - inside one process no collective is needed.
//   - With one process per GPU the ranks exchange exactly those three words: tcrc_rccl_allreduce_counts issues
//     ncclAllReduce(sum) over {mismatches, checked} and ncclAllReduce(min) over the first bad unit's global
//     index, on the caller's communicator and stream.  librccl is opened when the first of these calls is made
//     (the copy a framework has already loaded into the process, if any), so the library itself loads without it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "tcrc_internal.h"

using namespace tcrc_internal;

struct tcrc_group {
  std::vector<tcrc_ctx*> ctx;
  std::vector<int> device;
  std::vector<hipStream_t> stream;  // one per member, on the member's GPU
  std::vector<uint64_t*> d_counts;  // one TCRC_CNT_* block per member
};

namespace {

// Runs fn(member) on one thread per member and returns the first non-OK code (with that thread's error text made
// this thread's).
template <typename F>
int for_each_member(tcrc_group* g, F&& fn) {
  const size_t n = g->ctx.size();
  std::vector<int> rc(n, TCRC_OK);
  std::vector<std::string> err(n);
  auto run = [&](size_t i) {
    rc[i] = fn(i);
    if (rc[i] != TCRC_OK) err[i] = tcrc_last_error();
  };
  std::vector<std::thread> th;
  for (size_t i = 1; i < n; ++i) th.emplace_back(run, i);
  run(0);
  for (auto& t : th) t.join();
  for (size_t i = 0; i < n; ++i)
    if (rc[i] != TCRC_OK) return fail(rc[i], "group member " + std::to_string(i) + ": " + err[i]);
  return TCRC_OK;
}

}  // namespace

extern "C" {

void tcrc_shard_range(uint64_t n_units, uint32_t index, uint32_t count, uint64_t* lo, uint64_t* hi) {
  uint64_t a = 0, b = 0;
  if (count != 0 && index < count) {
    const uint64_t base = n_units / count, extra = n_units % count;
    a = index * base + (index < extra ? index : extra);
    b = a + base + (index < extra ? 1 : 0);
  }
  if (lo) *lo = a;
  if (hi) *hi = b;
}

int tcrc_group_shutdown(tcrc_group* g) {
  if (!g) return TCRC_OK;
  for (size_t i = 0; i < g->ctx.size(); ++i) {
    if (g->ctx[i]) {
      DeviceScope scope(g->device[i]);
      if (i < g->stream.size() && g->stream[i]) (void)hipStreamDestroy(g->stream[i]);
      if (i < g->d_counts.size() && g->d_counts[i]) (void)hipFree(g->d_counts[i]);
      (void)tcrc_shutdown(g->ctx[i]);
    }
  }
  delete g;
  return TCRC_OK;
}

int tcrc_group_init(const int* devices, int n, tcrc_group** out) {
  if (!out) return fail(TCRC_ERR_INVALID, "null out pointer");
  *out = nullptr;
  if (!devices || n <= 0 || n > 64) return fail(TCRC_ERR_INVALID, "a group has 1 to 64 members");
  tcrc_group* g = new tcrc_group();
  for (int i = 0; i < n; ++i) {
    tcrc_ctx* c = nullptr;
    const int rc = tcrc_init(devices[i], &c);
    if (rc != TCRC_OK) {
      const std::string why = tcrc_last_error();
      tcrc_group_shutdown(g);
      return fail(rc, "member " + std::to_string(i) + " (GPU " + std::to_string(devices[i]) + "): " + why);
    }
    g->ctx.push_back(c);
    g->device.push_back(devices[i]);
    g->stream.push_back(nullptr);
    g->d_counts.push_back(nullptr);
    DeviceScope scope(devices[i]);
    hipError_t e = scope.err;
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&g->stream[i], hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc((void**)&g->d_counts[i], 64);
    if (e != hipSuccess) {
      tcrc_group_shutdown(g);
      return fail(TCRC_ERR_HIP, std::string("group set-up: ") + hipGetErrorString(e));
    }
  }
  *out = g;
  return TCRC_OK;
}

int tcrc_group_size(const tcrc_group* g) { return g ? (int)g->ctx.size() : 0; }
tcrc_ctx* tcrc_group_ctx(tcrc_group* g, int member) {
  return (g && member >= 0 && (size_t)member < g->ctx.size()) ? g->ctx[member] : nullptr;
}
int tcrc_group_device(const tcrc_group* g, int member) {
  return (g && member >= 0 && (size_t)member < g->device.size()) ? g->device[member] : -1;
}

// Uniform blocks already resident on the members' GPUs: member m holds the blocks
// [lo_m, hi_m) = tcrc_shard_range(n_blocks, m, members) of the corpus at d_bases[m] (block lo_m first).
int tcrc_group_dev_blocks_verify(tcrc_group* g, const void* const* d_bases, uint64_t n_blocks, uint32_t block_len,
                                 uint64_t stride, uint32_t init, int suffix_byte, uint32_t flags,
                                 const uint32_t* const* d_expected, uint64_t* h_counts) {
  if (!g || !h_counts) return fail(TCRC_ERR_INVALID, "null argument");
  h_counts[TCRC_CNT_MISMATCH] = 0;
  h_counts[TCRC_CNT_FIRST_BAD] = ~0ull;
  h_counts[TCRC_CNT_CHECKED] = 0;
  if (n_blocks == 0) return TCRC_OK;
  if (!d_bases) return fail(TCRC_ERR_INVALID, "null argument");
  const uint32_t members = (uint32_t)g->ctx.size();
  std::vector<uint64_t> part((size_t)members * TCRC_CNT_WORDS);
  const int rc = for_each_member(g, [&](size_t m) -> int {
    uint64_t lo = 0, hi = 0;
    tcrc_shard_range(n_blocks, (uint32_t)m, members, &lo, &hi);
    uint64_t* mine = &part[m * TCRC_CNT_WORDS];
    mine[TCRC_CNT_MISMATCH] = 0;
    mine[TCRC_CNT_FIRST_BAD] = ~0ull;
    mine[TCRC_CNT_CHECKED] = 0;
    if (hi == lo) return TCRC_OK;
    if (!d_bases[m]) return fail(TCRC_ERR_INVALID, "null shard pointer");
    tcrc_ctx* c = g->ctx[m];
    DeviceScope scope(g->device[m]);
    if (scope.err != hipSuccess) return fail(TCRC_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(scope.err));
    if (int r = tcrc_dev_counts_reset(c, g->d_counts[m], g->stream[m])) return r;
    if (int r = tcrc_dev_blocks_verify(c, d_bases[m], hi - lo, block_len, stride, init, suffix_byte, nullptr, flags,
                                       d_expected ? d_expected[m] : nullptr, nullptr, nullptr, g->d_counts[m],
                                       g->stream[m]))
      return r;
    HIP_TRY(hipMemcpyAsync(mine, g->d_counts[m], TCRC_CNT_WORDS * 8, hipMemcpyDeviceToHost, g->stream[m]));
    HIP_TRY(hipStreamSynchronize(g->stream[m]));
    if (mine[TCRC_CNT_FIRST_BAD] != ~0ull) mine[TCRC_CNT_FIRST_BAD] += lo;  // the corpus's numbering
    return TCRC_OK;
  });
  if (rc != TCRC_OK) return rc;
  for (uint32_t m = 0; m < members; ++m) {
    const uint64_t* mine = &part[(size_t)m * TCRC_CNT_WORDS];
    h_counts[TCRC_CNT_MISMATCH] += mine[TCRC_CNT_MISMATCH];
    h_counts[TCRC_CNT_CHECKED] += mine[TCRC_CNT_CHECKED];
    if (mine[TCRC_CNT_FIRST_BAD] < h_counts[TCRC_CNT_FIRST_BAD]) h_counts[TCRC_CNT_FIRST_BAD] = mine[TCRC_CNT_FIRST_BAD];
  }
  return TCRC_OK;
}

// The same for blocks in host memory: every member uploads and verifies its range of the one image.
int tcrc_group_host_blocks_verify(tcrc_group* g, const void* h_base, uint64_t n_blocks, uint32_t block_len,
                                  uint64_t stride, uint32_t init, int suffix_byte, uint32_t flags,
                                  const uint32_t* h_expected, uint8_t* h_ok, uint64_t* h_counts) {
  if (!g || !h_counts) return fail(TCRC_ERR_INVALID, "null argument");
  h_counts[TCRC_CNT_MISMATCH] = 0;
  h_counts[TCRC_CNT_FIRST_BAD] = ~0ull;
  h_counts[TCRC_CNT_CHECKED] = 0;
  if (n_blocks == 0) return TCRC_OK;
  if (!h_base) return fail(TCRC_ERR_INVALID, "null argument");
  if (n_blocks > 1 && stride < block_len) return fail(TCRC_ERR_INVALID, "stride smaller than block_len");
  const uint32_t members = (uint32_t)g->ctx.size();
  std::vector<uint64_t> part((size_t)members * TCRC_CNT_WORDS);
  const int rc = for_each_member(g, [&](size_t m) -> int {
    uint64_t lo = 0, hi = 0;
    tcrc_shard_range(n_blocks, (uint32_t)m, members, &lo, &hi);
    uint64_t* mine = &part[m * TCRC_CNT_WORDS];
    mine[TCRC_CNT_MISMATCH] = 0;
    mine[TCRC_CNT_FIRST_BAD] = ~0ull;
    mine[TCRC_CNT_CHECKED] = 0;
    if (hi == lo) return TCRC_OK;
    if (int r = tcrc_host_blocks_verify(g->ctx[m], (const uint8_t*)h_base + lo * stride, hi - lo, block_len, stride, init,
                                        suffix_byte, flags, h_expected ? h_expected + lo : nullptr, nullptr,
                                        h_ok ? h_ok + lo : nullptr, mine))
      return r;
    if (mine[TCRC_CNT_FIRST_BAD] != ~0ull) mine[TCRC_CNT_FIRST_BAD] += lo;
    return TCRC_OK;
  });
  if (rc != TCRC_OK) return rc;
  for (uint32_t m = 0; m < members; ++m) {
    const uint64_t* mine = &part[(size_t)m * TCRC_CNT_WORDS];
    h_counts[TCRC_CNT_MISMATCH] += mine[TCRC_CNT_MISMATCH];
    h_counts[TCRC_CNT_CHECKED] += mine[TCRC_CNT_CHECKED];
    if (mine[TCRC_CNT_FIRST_BAD] < h_counts[TCRC_CNT_FIRST_BAD]) h_counts[TCRC_CNT_FIRST_BAD] = mine[TCRC_CNT_FIRST_BAD];
  }
  return TCRC_OK;
}

}  // extern "C"

// ---- RCCL ---------------------------------------------------------------------------------------------
namespace {

// the few entry points of rccl.h this file uses, with the types spelled out (ncclComm_t is an opaque pointer,
// ncclResult_t an int with 0 = success, ncclUniqueId 128 bytes)
struct Rccl {
  void* handle = nullptr;
  int (*get_unique_id)(void*) = nullptr;
  int (*comm_init_rank)(void**, int, const void*, int) = nullptr;  // (ncclComm_t*, nranks, ncclUniqueId by value, rank)
  int (*comm_destroy)(void*) = nullptr;
  int (*all_reduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  const char* (*error_string)(int) = nullptr;
  std::string why;
};
constexpr int kNcclUint64 = 5, kNcclSum = 0, kNcclMin = 3;  // ncclUint64, ncclSum, ncclMin (rccl.h)

Rccl* rccl() {
  static std::mutex mu;
  static Rccl r;
  std::lock_guard<std::mutex> lock(mu);
  if (r.handle || !r.why.empty()) return &r;
  for (const char* name : {"librccl.so.1", "librccl.so"}) {
    r.handle = dlopen(name, RTLD_NOW | RTLD_NOLOAD);  // the copy already in the process (a framework's), if any
    if (r.handle) break;
  }
  for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
    if (r.handle) break;
    r.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
  }
  if (!r.handle) {
    r.why = std::string("librccl could not be opened: ") + dlerror();
    return &r;
  }
  r.get_unique_id = (int (*)(void*))dlsym(r.handle, "ncclGetUniqueId");
  r.comm_destroy = (int (*)(void*))dlsym(r.handle, "ncclCommDestroy");
  r.all_reduce = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))dlsym(r.handle, "ncclAllReduce");
  r.error_string = (const char* (*)(int))dlsym(r.handle, "ncclGetErrorString");
  r.comm_init_rank = (int (*)(void**, int, const void*, int))dlsym(r.handle, "ncclCommInitRank");
  if (!r.get_unique_id || !r.comm_destroy || !r.all_reduce || !r.comm_init_rank) {
    r.why = "librccl lacks ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllReduce";
    r.handle = nullptr;
  }
  return &r;
}

int rccl_fail(Rccl* r, const char* what, int code) {
  return fail(TCRC_ERR_HIP, std::string(what) + ": " + (r->error_string ? r->error_string(code) : "RCCL error") + " (" +
                                std::to_string(code) + ")");
}

// Before the exchange: first_bad from the rank's own numbering to the corpus's (~0 stays ~0), and the two sums
// next to each other ({mismatches, checked, first bad}) so that one all-reduce carries both.  After it: back to
// the layout of the counter block.
__global__ void k_counts_pack(unsigned long long* c, unsigned long long first_unit) {
  unsigned long long fb = c[TCRC_CNT_FIRST_BAD];
  if (fb != ~0ull) fb += first_unit;
  const unsigned long long checked = c[TCRC_CNT_CHECKED];
  c[1] = checked;
  c[2] = fb;
}
__global__ void k_counts_unpack(unsigned long long* c) {
  const unsigned long long checked = c[1], fb = c[2];
  c[TCRC_CNT_FIRST_BAD] = fb;
  c[TCRC_CNT_CHECKED] = checked;
}

struct UniqueId {
  char bytes[128];
};

}  // namespace

extern "C" {

int tcrc_rccl_get_unique_id(void* id128) {
  if (!id128) return fail(TCRC_ERR_INVALID, "null argument");
  Rccl* r = rccl();
  if (!r->handle) return fail(TCRC_ERR_NO_DEVICE, r->why);
  if (int e = r->get_unique_id(id128)) return rccl_fail(r, "ncclGetUniqueId", e);
  return TCRC_OK;
}

int tcrc_rccl_comm_init_rank(tcrc_ctx* ctx, int nranks, const void* id128, int rank, void** comm_out) {
  if (!ctx || !id128 || !comm_out) return fail(TCRC_ERR_INVALID, "null argument");
  *comm_out = nullptr;
  Rccl* r = rccl();
  if (!r->handle) return fail(TCRC_ERR_NO_DEVICE, r->why);
  // the communicator belongs to the GPU that is current when it is made; the caller's current GPU is left as it
  // is afterwards (ncclUniqueId is passed by value: a 128-byte struct)
  TCRC_DEVICE_SCOPE(ctx);
  UniqueId id;
  __builtin_memcpy(id.bytes, id128, sizeof(id.bytes));
  typedef int (*init_fn)(void**, int, UniqueId, int);
  if (int e = ((init_fn)(void*)r->comm_init_rank)(comm_out, nranks, id, rank)) return rccl_fail(r, "ncclCommInitRank", e);
  return TCRC_OK;
}

int tcrc_rccl_comm_destroy(void* comm) {
  if (!comm) return TCRC_OK;
  Rccl* r = rccl();
  if (!r->handle) return fail(TCRC_ERR_NO_DEVICE, r->why);
  if (int e = r->comm_destroy(comm)) return rccl_fail(r, "ncclCommDestroy", e);
  return TCRC_OK;
}

int tcrc_rccl_allreduce_counts(tcrc_ctx* ctx, void* comm, uint64_t* d_counts, uint64_t first_unit_of_rank, void* stream) {
  if (!ctx || !comm || !d_counts) return fail(TCRC_ERR_INVALID, "null argument");
  Rccl* r = rccl();
  if (!r->handle) return fail(TCRC_ERR_NO_DEVICE, r->why);
  TCRC_DEVICE_SCOPE(ctx);
  hipStream_t s = (hipStream_t)stream;
  static_assert(TCRC_CNT_MISMATCH == 0 && TCRC_CNT_FIRST_BAD == 1 && TCRC_CNT_CHECKED == 2, "layout of the counter block");
  hipLaunchKernelGGL(k_counts_pack, dim3(1), dim3(1), 0, s, (unsigned long long*)d_counts,
                     (unsigned long long)first_unit_of_rank);
  HIP_TRY(hipGetLastError());
  // the sums of {mismatches, checked}, and the minimum of the first bad unit's index (~0 = none is the largest
  // value, so a clean rank never wins)
  if (int e = r->all_reduce(d_counts, d_counts, 2, kNcclUint64, kNcclSum, comm, s)) return rccl_fail(r, "ncclAllReduce(sum)", e);
  if (int e = r->all_reduce(d_counts + 2, d_counts + 2, 1, kNcclUint64, kNcclMin, comm, s))
    return rccl_fail(r, "ncclAllReduce(min)", e);
  hipLaunchKernelGGL(k_counts_unpack, dim3(1), dim3(1), 0, s, (unsigned long long*)d_counts);
  HIP_TRY(hipGetLastError());
  return TCRC_OK;
}

}  // extern "C"
