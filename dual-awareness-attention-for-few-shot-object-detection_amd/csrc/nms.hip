// Greedy NMS entirely on device (no D2H mask copy, no host scan) for gfx950.
//
// Replaces the reference's lib/model/csrc/cuda/nms.cu:
//   nms_kernel (:23-67)   -> nms_mask_kernel : 64x64 IoU tiles -> 64-bit suppression words, upper
//                            triangle only (the lower one is never read); one wave per tile, the
//                            grid enumerates (column group, row block) so every workgroup has the
//                            same amount of work (the reference's row-major sweep is triangular).
//   host scan (:100-123)  -> nms_scan_kernel : one 1024-thread workgroup per problem walks the
//                            64-box blocks in order. Wave 0 (the resolver) settles block b with
//                            scalar bit tricks (s_ff1 + v_readlane) and ORs the kept rows' words
//                            for block b+1 from registers it prefetched one iteration earlier;
//                            waves 1..15 (the workers) OR the kept rows of block b-1 into the
//                            LDS-resident `remv` for every later block, one iteration behind, with
//                            unconditional coalesced loads. One barrier per block; nothing on the
//                            resolver's critical path waits for HBM.
// IoU uses the legacy "+1" widths (nms.cu:13-21). `inclusive`==0 suppresses on IoU > thr
// (reference CUDA, nms.cu:60); ==1 on IoU >= thr (reference CPU, cpu/nms_cpu.cpp:60).
// Compiled with -ffp-contract=off. The division-free fast path below only decides cases that are
// at least 2^-21 (relative) away from the threshold; anything closer takes the exact IEEE division,
// so every decision equals the reference's `inter / (Sa + Sb - inter) > thr` bit for bit.
#include "common.h"
#include "../../include/dana_hip.h"

namespace {

__device__ __forceinline__ bool iou_suppress(float4 a, float sa, float4 b, float sb, float thr, int inclusive) {
  const float left = fmaxf(a.x, b.x), right = fminf(a.z, b.z);
  const float top = fmaxf(a.y, b.y), bottom = fminf(a.w, b.w);
  const float w = fmaxf(right - left + 1.f, 0.f), h = fmaxf(bottom - top + 1.f, 0.f);
  const float inter = w * h;
  const float uni = sa + sb - inter;
  const float t = thr * uni;
  if (uni > 0.f && thr > 0.f) {
    if (inter > t * (1.f + 4.8e-7f)) return true;   // certainly above the threshold
    if (inter < t * (1.f - 4.8e-7f)) return false;  // certainly below
  }
  const float ovr = inter / uni;
  return inclusive ? (ovr >= thr) : (ovr > thr);
}

__device__ __forceinline__ float box_area(float4 a) { return (a.z - a.x + 1.f) * (a.w - a.y + 1.f); }

// grid = (ceil(col_blocks/4), row_blocks, problems); wave w of the workgroup owns tile (rb, cg*4 + w)
__global__ void __launch_bounds__(256)
nms_mask_kernel(const float4* __restrict__ boxes, unsigned long long* __restrict__ mask, int n, int col_blocks,
                float thr, int inclusive) {
  __shared__ float4 colbox[4][64];
  __shared__ float colarea[4][64];
  const int rb = blockIdx.y;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int cb = blockIdx.x * 4 + wave;
  if (cb < rb || cb >= col_blocks) return;  // wave-uniform: whole waves leave, no barrier below
  const float4* pb = boxes + (long)blockIdx.z * n;
  unsigned long long* pm = mask + (long)blockIdx.z * n * col_blocks;
  const int row = rb * 64 + lane;
  const float4 me = row < n ? pb[row] : make_float4(0, 0, 0, 0);
  const float sme = box_area(me);
  const int col = cb * 64 + lane;
  const float4 cbx = col < n ? pb[col] : make_float4(0, 0, 0, 0);
  colbox[wave][lane] = cbx;
  colarea[wave][lane] = box_area(cbx);
  __builtin_amdgcn_wave_barrier();
  const int csize = min(64, n - cb * 64);
  const int start = (cb == rb) ? lane + 1 : 0;
  unsigned long long t = 0;
  for (int i = 0; i < csize; ++i)
    if (i >= start && iou_suppress(me, sme, colbox[wave][i], colarea[wave][i], thr, inclusive)) t |= 1ULL << i;
  if (row < n) pm[(long)row * col_blocks + cb] = t;
}

__device__ __forceinline__ unsigned long long readlane64(unsigned long long v, int l) {
  unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)v, l);
  unsigned hi = __builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l);
  return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long uniform64(unsigned long long v) {
  return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32)) << 32) |
         (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
}
__device__ __forceinline__ unsigned long long wave_or64(unsigned long long v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v |= __shfl_xor(v, o);
  return v;
}

// LDS: remv[col_blocks] | list[2][64] ints | cnt[2] | count | done
__global__ void __launch_bounds__(1024)
nms_scan_kernel(const unsigned long long* __restrict__ mask, int n, int col_blocks, int max_keep,
                int* __restrict__ keep, int* __restrict__ num_keep, int keep_stride) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long remv[];
  int* list = (int*)(remv + col_blocks);  // [2][64] kept rows (local index) of the last two blocks
  int* cnt = list + 128;                  // [2]
  int* s_count = cnt + 2;
  int* s_done = cnt + 3;
  const unsigned long long* pm = mask + (long)blockIdx.x * n * col_blocks;
  int* pk = keep + (long)blockIdx.x * keep_stride;
  for (int j = threadIdx.x; j < col_blocks; j += blockDim.x) remv[j] = 0;
  if (threadIdx.x < 4) cnt[threadIdx.x] = 0;  // cnt[0..1], count, done
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // resolver prefetch registers: diagonal word and the word for the next block, rows of block b
  unsigned long long diag = 0, nxt = 0;
  if (wave == 0) {
    if (lane < n) {
      diag = pm[(long)lane * col_blocks];
      if (col_blocks > 1) nxt = pm[(long)lane * col_blocks + 1];
    }
  }
  for (int b = 0; b < col_blocks; ++b) {
    if (wave == 0) {
      // ---- resolver: prefetch block b+1's words first so their latency hides under the resolve ----
      unsigned long long diag_n = 0, nxt_n = 0;
      const int rown = (b + 1) * 64 + lane;
      if (b + 1 < col_blocks && rown < n) {
        diag_n = pm[(long)rown * col_blocks + b + 1];
        if (b + 2 < col_blocks) nxt_n = pm[(long)rown * col_blocks + b + 2];
      }
      const int bsize = min(64, n - b * 64);
      const unsigned long long valid = bsize == 64 ? ~0ULL : ((1ULL << bsize) - 1);
      unsigned long long alive = uniform64(~remv[b] & valid);
      unsigned long long kept = 0, fast = 0;  // fast: the kept rows' words for block b+1, gathered as they are kept
      const int base = *s_count;
      int count = base;
      while (alive) {
        const int k = __builtin_ctzll(alive);
        kept |= 1ULL << k;
        alive &= ~(1ULL << k);
        alive &= ~readlane64(diag, k);
        fast |= readlane64(nxt, k);
        if (++count == max_keep) break;
      }
      const bool mine = (kept >> lane) & 1ULL;
      const int pos = __builtin_popcountll(kept & ((1ULL << lane) - 1));
      if (mine) {
        pk[base + pos] = b * 64 + lane;
        list[(b & 1) * 64 + pos] = lane;
      }
      if (lane == 0) {
        if (b + 1 < col_blocks && fast) atomicOr(&remv[b + 1], fast);
        cnt[b & 1] = count - base;
        *s_count = count;
        if (count == max_keep) *s_done = 1;
      }
      diag = diag_n;
      nxt = nxt_n;
    } else if (b > 0) {
      // ---- workers: rows kept in block b-1 -> remv[j], j >= b+1 (one iteration behind the resolver) ----
      // The (kept row, group of 64 column words) items are dealt round-robin to the 15 worker waves and each wave issues
      // ALL its loads before it uses any (up to 6 in flight): one L2 round trip per block instead of one per column
      // group (round 1: 3.2 us per block; now 2.5, the resolver wave's chain being what is left).
      const int c = cnt[(b - 1) & 1];
      const int* rows = list + ((b - 1) & 1) * 64;
      const unsigned long long* base = pm + (long)(b - 1) * 64 * col_blocks;
      const int first = b + 1;
      const int ngroups = (col_blocks - first + 63) / 64;
      const int items = c * ngroups;
      constexpr int U = 6;
      for (int it0 = wave - 1; it0 < items; it0 += 15 * U) {
        unsigned long long v[U];
        int jj[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int it = it0 + u * 15;
          v[u] = 0;
          jj[u] = -1;
          if (it < items) {
            const int i = it / ngroups, g = it - i * ngroups;
            const int j = first + g * 64 + lane;
            if (j < col_blocks) {
              jj[u] = j;
              v[u] = base[(long)rows[i] * col_blocks + j];
            }
          }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (jj[u] >= 0 && v[u]) atomicOr(&remv[jj[u]], v[u]);
      }
    }
    __syncthreads();
    if (*s_done) break;
  }
  if (threadIdx.x == 0) num_keep[blockIdx.x] = *s_count;
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
  const size_t lds = (size_t)cb * 8 + 128 * 4 + 16;
  DANA_CHECK_ARG(lds <= 64 * 1024 && cb <= 65535, "dana_nms: n=%d too large for the LDS-resident scan", n);
  dim3 grid((cb + 3) / 4, cb, problems);
  nms_mask_kernel<<<grid, 256, 0, s>>>((const float4*)boxes, (unsigned long long*)workspace, n, cb, thr, inclusive);
  DANA_CHECK_LAUNCH("dana_nms(mask)");
  nms_scan_kernel<<<problems, 1024, lds, s>>>((const unsigned long long*)workspace, n, cb, max_keep, keep, num_keep,
                                              keep_stride);
  DANA_CHECK_LAUNCH("dana_nms(scan)");
  return DANA_OK;
}

}  // extern "C"
