// Development aid: what one "launch a kernel and wait for it" costs on this box, by the way the host waits.
//   hipcc --offload-arch=gfx950 -O2 -Iinclude tools/floor_probe.hip -Lterarkdb_amd/lib -ltcrc -o gpurun_out/floor_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <utility>
#include <vector>
#include "tcrc.h"

__global__ void k_empty() {}
__global__ void k_flag(volatile uint32_t* flag, uint32_t v) {
  if (threadIdx.x == 0) __hip_atomic_store((uint32_t*)flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

static double now_us() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main() {
  hipStream_t s;
  hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  uint32_t* h_flag;
  hipHostMalloc((void**)&h_flag, 64, hipHostMallocMapped | hipHostMallocCoherent);
  *h_flag = 0;
  uint32_t* d_flag;
  hipHostGetDevicePointer((void**)&d_flag, h_flag, 0);
  const int K = 2000;
  for (int i = 0; i < 200; ++i) { k_empty<<<1, 64, 0, s>>>(); hipStreamSynchronize(s); }
  double t0 = now_us();
  for (int i = 0; i < K; ++i) { k_empty<<<1, 64, 0, s>>>(); hipStreamSynchronize(s); }
  printf("empty kernel + hipStreamSynchronize: %.2f us\n", (now_us() - t0) / K);
  t0 = now_us();
  for (int i = 0; i < K; ++i) { k_empty<<<256, 1024, 0, s>>>(); hipStreamSynchronize(s); }
  printf("empty 256x1024 kernel + hipStreamSynchronize: %.2f us\n", (now_us() - t0) / K);
  t0 = now_us();
  for (int i = 1; i <= K; ++i) {
    k_flag<<<1, 64, 0, s>>>(d_flag, (uint32_t)i);
    while (__atomic_load_n(h_flag, __ATOMIC_ACQUIRE) != (uint32_t)i) {}
  }
  printf("flag kernel + poll on pinned word: %.2f us\n", (now_us() - t0) / K);
  hipStreamSynchronize(s);
  hipEvent_t ev;
  hipEventCreateWithFlags(&ev, hipEventDisableTiming);
  t0 = now_us();
  for (int i = 0; i < K; ++i) {
    k_empty<<<1, 64, 0, s>>>();
    hipEventRecord(ev, s);
    while (hipEventQuery(ev) == hipErrorNotReady) {}
  }
  printf("empty kernel + event record + hipEventQuery spin: %.2f us\n", (now_us() - t0) / K);
  t0 = now_us();
  for (int i = 0; i < K; ++i) {
    k_empty<<<1, 64, 0, s>>>();
    while (hipStreamQuery(s) == hipErrorNotReady) {}
  }
  printf("empty kernel + hipStreamQuery spin: %.2f us\n", (now_us() - t0) / K);

  tcrc_ctx* ctx = nullptr;
  if (tcrc_init(0, &ctx) != TCRC_OK) { printf("init failed: %s\n", tcrc_last_error()); return 1; }
  const size_t total = 64u << 22;
  uint8_t* d_data;
  hipMalloc((void**)&d_data, total);
  hipMemset(d_data, 0x5a, total);
  uint32_t* d_out;
  hipMalloc((void**)&d_out, 64 * 4);
  for (size_t sz : {size_t(256) << 10, size_t(1) << 20, size_t(4) << 20}) {
    for (int n : {1, 64}) {
      std::vector<tcrc_desc> d(n);
      for (int i = 0; i < n; ++i) { d[i].offset = i * sz; d[i].len = (uint32_t)sz; d[i].init = 0; }
      for (int i = 0; i < 300; ++i) tcrc_dev_extend_long(ctx, d_data, d.data(), n, 0, d_out, s);
      t0 = now_us();
      for (int i = 0; i < K; ++i) tcrc_dev_extend_long(ctx, d_data, d.data(), n, 0, d_out, s);
      double us = (now_us() - t0) / K;
      printf("tcrc_dev_extend_long %2d x %7zu: %.2f us  %.1f GB/s\n", n, sz, us, n * sz / us / 1e3);
      t0 = now_us();
      for (int i = 0; i < K; ++i) { tcrc_dev_extend_long_noalloc(ctx, d_data, d.data(), n, 0, d_out, s); }
      hipStreamSynchronize(s);
      us = (now_us() - t0) / K;
      printf("   noalloc, back to back (no wait per call): %.2f us\n", us);
    }
  }
  for (auto shape : {std::pair<int, size_t>(1, size_t(256) << 20), std::pair<int, size_t>(4, size_t(64) << 20),
                     std::pair<int, size_t>(64, size_t(4) << 20), std::pair<int, size_t>(1, size_t(64) << 20),
                     std::pair<int, size_t>(16, size_t(1) << 20)}) {
    const int n = shape.first;
    const size_t sz = shape.second;
    std::vector<tcrc_desc> d(n);
    for (int i = 0; i < n; ++i) { d[i].offset = i * sz; d[i].len = (uint32_t)sz; d[i].init = 0; }
    for (uint32_t seg : {0u}) {
      for (int i = 0; i < 300; ++i) tcrc_dev_extend_long(ctx, d_data, d.data(), n, seg, d_out, s);
      t0 = now_us();
      for (int i = 0; i < K; ++i) tcrc_dev_extend_long(ctx, d_data, d.data(), n, seg, d_out, s);
      const double us = (now_us() - t0) / K;
      printf("tcrc_dev_extend_long %2d x %9zu seg %5u: %.2f us  %.1f GB/s\n", n, sz, seg, us, n * sz / us / 1e3);
      if (seg == 0) {
        t0 = now_us();
        for (int i = 0; i < K; ++i) tcrc_dev_extend_long_noalloc(ctx, d_data, d.data(), n, seg, d_out, s);
        hipStreamSynchronize(s);
        printf("   noalloc, back to back (no wait per call): %.2f us\n", (now_us() - t0) / K);
      }
    }
  }
  tcrc_shutdown(ctx);
  return 0;
}
