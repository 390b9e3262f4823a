// Greedy NMS entirely on device (no D2H mask copy, no host scan) for gfx950.
//
// Replaces the reference's lib/model/csrc/cuda/nms.cu:
//   nms_kernel (:23-67)  -> nms_mask_kernel : 64x64 IoU tiles -> 64-bit suppression words,
//                            upper triangle only (the lower one is never read)
//   host scan  (:100-123) -> nms_scan_kernel : one workgroup per problem walks the 64-box blocks
//                            in order; wave 0 resolves the in-block dependency chain with
//                            scalar bit tricks (ctz + v_readlane), then all lanes OR the kept
//                            rows' mask words into the LDS-resident `remv` vector.
// IoU uses the legacy "+1" widths (nms.cu:13-21). `inclusive`==0 suppresses on IoU > thr
// (reference CUDA, nms.cu:60); ==1 on IoU >= thr (reference CPU, cpu/nms_cpu.cpp:60).
// Compiled with -ffp-contract=off: the IoU rounds exactly like oracle/dana_oracle.c.
#include "common.h"
#include "../../include/dana_hip.h"

namespace {

__device__ __forceinline__ bool iou_suppress(float4 a, float4 b, float thr, int inclusive) {
  float left = fmaxf(a.x, b.x), right = fminf(a.z, b.z);
  float top = fmaxf(a.y, b.y), bottom = fminf(a.w, b.w);
  float w = fmaxf(right - left + 1.f, 0.f), h = fmaxf(bottom - top + 1.f, 0.f);
  float inter = w * h;
  float sa = (a.z - a.x + 1.f) * (a.w - a.y + 1.f);
  float sb = (b.z - b.x + 1.f) * (b.w - b.y + 1.f);
  float ovr = inter / (sa + sb - inter);
  return inclusive ? (ovr >= thr) : (ovr > thr);
}

// grid = (row_blocks, problems); 256 threads = 4 waves; wave w sweeps column blocks rb+w, rb+w+4, ...
__global__ void __launch_bounds__(256)
nms_mask_kernel(const float4* __restrict__ boxes, unsigned long long* __restrict__ mask, int n, int col_blocks,
                float thr, int inclusive) {
  __shared__ float4 colbox[4][64];
  const int rb = blockIdx.x;
  const float4* pb = boxes + (long)blockIdx.y * n;
  unsigned long long* pm = mask + (long)blockIdx.y * n * col_blocks;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = rb * 64 + lane;
  const float4 me = row < n ? pb[row] : make_float4(0, 0, 0, 0);
  for (int cb = rb + wave; cb < col_blocks; cb += 4) {
    const int col = cb * 64 + lane;
    colbox[wave][lane] = col < n ? pb[col] : make_float4(0, 0, 0, 0);
    __builtin_amdgcn_wave_barrier();
    const int csize = min(64, n - cb * 64);
    const int start = (cb == rb) ? lane + 1 : 0;
    unsigned long long t = 0;
    for (int i = start; i < csize; ++i)
      if (iou_suppress(me, colbox[wave][i], thr, inclusive)) t |= 1ULL << i;
    if (row < n) pm[(long)row * col_blocks + cb] = t;
    __builtin_amdgcn_wave_barrier();
  }
}

__device__ __forceinline__ unsigned long long readlane64(unsigned long long v, int l) {
  unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)v, l);
  unsigned hi = __builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l);
  return ((unsigned long long)hi << 32) | lo;
}

// grid = problems; 1024 threads. Emits kept positions (ascending) and their count.
__global__ void __launch_bounds__(1024)
nms_scan_kernel(const unsigned long long* __restrict__ mask, int n, int col_blocks, int max_keep,
                int* __restrict__ keep, int* __restrict__ num_keep, int keep_stride) {
  // all LDS in ONE dynamic array (16-B aligned base): remv[col_blocks] | kept word | count word
  extern __shared__ __attribute__((aligned(16))) unsigned long long remv[];
  unsigned long long& s_kept = remv[col_blocks];
  int& s_count = *(int*)&remv[col_blocks + 1];
  const unsigned long long* pm = mask + (long)blockIdx.x * n * col_blocks;
  int* pk = keep + (long)blockIdx.x * keep_stride;
  for (int j = threadIdx.x; j < col_blocks; j += blockDim.x) remv[j] = 0;
  if (threadIdx.x == 0) s_count = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  for (int b = 0; b < col_blocks; ++b) {
    if (threadIdx.x < 64) {
      const int row = b * 64 + lane;
      const unsigned long long diag = row < n ? pm[(long)row * col_blocks + b] : 0ULL;
      const int bsize = min(64, n - b * 64);
      const unsigned long long valid = bsize == 64 ? ~0ULL : ((1ULL << bsize) - 1);
      unsigned long long alive = ~remv[b] & valid;  // wave-uniform: keep it provably scalar
      alive = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(alive >> 32)) << 32) |
              (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)alive);
      unsigned long long kept = 0;
      const int base = s_count;
      int count = base;
      while (alive) {
        const int k = __builtin_ctzll(alive);
        kept |= 1ULL << k;
        alive &= ~(1ULL << k);
        alive &= ~readlane64(diag, k);
        if (++count == max_keep) break;
      }
      if ((kept >> lane) & 1ULL) {
        const int pos = base + __builtin_popcountll(kept & ((1ULL << lane) - 1));
        pk[pos] = b * 64 + lane;
      }
      if (lane == 0) {
        s_kept = kept;
        s_count = count;
      }
    }
    __syncthreads();
    const unsigned long long kept = s_kept;
    const bool done = (s_count == max_keep);
    if (done) break;
    if (kept) {
      for (int j = b + 1 + threadIdx.x; j < col_blocks; j += blockDim.x) {
        unsigned long long acc = 0;
        const unsigned long long* rowp = pm + (long)b * 64 * col_blocks + j;
#pragma unroll 8
        for (int k = 0; k < 64; ++k)
          if ((kept >> k) & 1ULL) acc |= rowp[(long)k * col_blocks];
        remv[j] |= acc;
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) num_keep[blockIdx.x] = s_count;
}

}  // namespace

extern "C" {

size_t dana_nms_workspace_bytes(int n, int problems) {
  if (n <= 0 || problems <= 0) return 0;
  size_t cb = (size_t)(n + 63) / 64;
  return (size_t)problems * n * cb * sizeof(unsigned long long);
}

int dana_nms(const float* boxes, int n, int problems, float thr, int inclusive, int max_keep, int* keep,
             int keep_stride, int* num_keep, void* workspace, size_t workspace_bytes, dana_stream_t stream) {
  DANA_CHECK_ARG(n >= 0 && problems >= 0, "dana_nms: bad n=%d problems=%d", n, problems);
  hipStream_t s = (hipStream_t)stream;
  if (problems == 0) return DANA_OK;
  DANA_CHECK_ARG(num_keep, "dana_nms: null num_keep");
  if (n == 0) {  // nms.h:17-18: empty input -> empty result
    if (hipMemsetAsync(num_keep, 0, sizeof(int) * problems, s) != hipSuccess) {
      dana_set_error("dana_nms: memset failed");
      return DANA_ERR_HIP;
    }
    return DANA_OK;
  }
  DANA_CHECK_ARG(boxes && keep, "dana_nms: null pointer");
  if (max_keep <= 0 || max_keep > n) max_keep = n;
  DANA_CHECK_ARG(keep_stride >= max_keep, "dana_nms: keep_stride %d < max_keep %d", keep_stride, max_keep);
  DANA_CHECK_ARG(((uintptr_t)boxes & 15) == 0, "dana_nms: boxes must be 16-byte aligned");
  size_t need = dana_nms_workspace_bytes(n, problems);
  if (workspace_bytes < need || !workspace) {
    dana_set_error("dana_nms: workspace %zu < %zu", workspace_bytes, need);
    return DANA_ERR_WORKSPACE;
  }
  const int cb = (n + 63) / 64;
  DANA_CHECK_ARG((size_t)(cb + 2) * 8 <= 64 * 1024, "dana_nms: n=%d too large for the LDS-resident scan", n);
  dim3 grid(cb, problems);
  nms_mask_kernel<<<grid, 256, 0, s>>>((const float4*)boxes, (unsigned long long*)workspace, n, cb, thr, inclusive);
  DANA_CHECK_LAUNCH("dana_nms(mask)");
  nms_scan_kernel<<<problems, 1024, (size_t)(cb + 2) * 8, s>>>((const unsigned long long*)workspace, n, cb, max_keep, keep,
                                                        num_keep, keep_stride);
  DANA_CHECK_LAUNCH("dana_nms(scan)");
  return DANA_OK;
}

}  // extern "C"
