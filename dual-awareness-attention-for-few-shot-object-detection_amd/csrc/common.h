// Shared helpers for the libdana_hip.so C-ABI layer (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <string.h>
#include <atomic>
#include <mutex>
#include <initializer_list>

#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "libdana_hip.so is gfx950-only: kernels here declare up to ~69 KB of static and 160 KB of dynamic LDS (CDNA4's 160 KB per CU) and use gfx950 MFMA / LDS-DMA instructions"
#endif

#define DANA_OK 0
#define DANA_ERR_ARG (-1)
#define DANA_ERR_HIP (-2)
#define DANA_ERR_WORKSPACE (-3)

// thread-local last-error text, read back through dana_last_error().
void dana_set_error(const char* fmt, ...);

#define DANA_CHECK_ARG(cond, ...)                 \
  do {                                            \
    if (!(cond)) {                                \
      dana_set_error(__VA_ARGS__);                \
      return DANA_ERR_ARG;                        \
    }                                             \
  } while (0)

#define DANA_CHECK_LAUNCH(name)                                               \
  do {                                                                        \
    hipError_t e__ = hipGetLastError();                                       \
    if (e__ != hipSuccess) {                                                  \
      dana_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));  \
      return DANA_ERR_HIP;                                                    \
    }                                                                         \
  } while (0)

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-DEVICE attribute of a kernel: the opt-in above 64 KiB is made
// once per (kernel, device) -- a process that drives several devices (nn.DataParallel's thread per device,
// include/dana_hip.h) launches on each of them. One static DeviceOnce per kernel instantiation.
struct DeviceOnce {
  std::atomic<unsigned long long> done{0};  // bit d: device d has the attribute (devices >= 64: set on every call)
  std::mutex mu;
  // Runs `set` (-> hipError_t) unless this device already has the attribute. The bit is published only AFTER a successful
  // call, under a mutex: a second host thread on the same device either sees the bit (the opt-in has happened) or waits
  // for it -- it can never launch in between -- and a failed opt-in is retried by the next launch.
  template <class F>
  void once(F&& set) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
      (void)set();
      return;
    }
    const unsigned long long bit = 1ull << dev;
    if (done.load(std::memory_order_acquire) & bit) return;
    std::lock_guard<std::mutex> g(mu);
    if (done.load(std::memory_order_relaxed) & bit) return;
    if (set() == hipSuccess) done.fetch_or(bit, std::memory_order_release);
  }
};

static inline int dana_ceil_div(long a, long b) { return (int)((a + b - 1) / b); }
static inline size_t dana_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// internal (wgrad.hip): batched "TN" GEMM out[z][N][K] = dY[z]^T . X[z] over M rows (K % 64 == 0, N % 4 == 0)
size_t dana_wgrad_tn_batched_workspace(int planes, int M, int N, int K);
int dana_wgrad_tn_batched(const float* dY, const float* X, float* out, int planes, int M, int N, int K, long batch_y,
                          long batch_x, void* workspace, size_t workspace_bytes, void* stream);

#ifdef __HIPCC__
// The exact three-way bf16 split of four fp32 values (igemm.hip: x = h + m + l, each a bf16 taken by truncation), packed
// as the planes store them: 4 bf16 = 8 bytes per plane. Shared by the contraction kernels' staging path, dana_split_weight
// and the producers that write an activation as split planes (winograd.hip).
__device__ __forceinline__ void dana_split3(const float4& v, uint2& h, uint2& m, uint2& l) {
  const float x[4] = {v.x, v.y, v.z, v.w};
  unsigned hb[4], mb[4], lb[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    hb[q] = __float_as_uint(x[q]) & 0xffff0000u;
    const float r1 = x[q] - __uint_as_float(hb[q]);  // exact
    mb[q] = __float_as_uint(r1) & 0xffff0000u;
    const float r2 = r1 - __uint_as_float(mb[q]);    // exact; <= 8 significant bits left
    lb[q] = __float_as_uint(r2);
  }
  // pack the upper halves of two dwords: bytes {S1.2, S1.3, S0.2, S0.3}
  h.x = __builtin_amdgcn_perm(hb[1], hb[0], 0x07060302u);
  h.y = __builtin_amdgcn_perm(hb[3], hb[2], 0x07060302u);
  m.x = __builtin_amdgcn_perm(mb[1], mb[0], 0x07060302u);
  m.y = __builtin_amdgcn_perm(mb[3], mb[2], 0x07060302u);
  l.x = __builtin_amdgcn_perm(lb[1], lb[0], 0x07060302u);
  l.y = __builtin_amdgcn_perm(lb[3], lb[2], 0x07060302u);
}
#endif
