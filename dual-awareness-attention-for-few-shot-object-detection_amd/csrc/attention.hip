// Non-GEMM pieces of the dual-awareness attention (BA block + CISA), gfx950.
// The contractions (Q/K projections, QK^T, A*S) run on dana_gemm_nt (igemm.hip); this file holds
// the wavefront-reduction kernels between them.
//
// Reference semantics replaced (lib/model/framework/dana.py, not code):
//   :134-137  BA block   w = softmax_pos(Linear(1024->1)(S)); g = w^T S; S += gamma * leaky_relu(g)
//   :144-145  unary      u = softmax_pos(Linear(1024->1)(S))                       (:276-277 at RoI level)
//   :143,146  A = softmax_keys(QK^T/16) + 0.1 * u^T                                (:274,278)
//   :150      mean over shots -- folded here by scaling A with 1/shot and letting the A*S GEMM run
//             over the concatenated keys of all shots (K = shot*L).
#include "common.h"
#include "../../include/dana_hip.h"
#include <float.h>

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// out[r] = x[r][:] . w + b ; one wave per row, float4 lanes
__global__ void __launch_bounds__(256)
rowdot_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
              float* __restrict__ out, long rows, int D, long ld) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float4* xr = (const float4*)(x + row * ld);
  const float4* wr = (const float4*)w;
  float s = 0.f;
  for (int c = lane; c < D / 4; c += 64) {
    const float4 a = xr[c], b = wr[c];
    s += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
  }
  s = wave_sum(s);
  if (lane == 0) out[row] = s + (bias ? bias[0] : 0.f);
}

// in-place softmax over the last dim of x[G][L] (ld = row stride); one wave per row
__global__ void __launch_bounds__(256)
softmax_rows_kernel(float* __restrict__ x, long G, int L, long ld, const float* __restrict__ src = nullptr, long ld_src = 0) {
  const int lane = threadIdx.x & 63;
  const long g = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (g >= G) return;
  float* r = x + g * ld;
  const float* in = src ? src + g * ld_src : r;  // (src: out-of-place, the scores stay as they are)
  float m = -FLT_MAX;
  for (int l = lane; l < L; l += 64) m = fmaxf(m, in[l]);
  m = wave_max(m);
  float s = 0.f;
  for (int l = lane; l < L; l += 64) {
    const float e = expf(in[l] - m);
    r[l] = e;
    s += e;
  }
  s = wave_sum(s);
  for (int l = lane; l < L; l += 64) r[l] = r[l] / s;
}

// BA block apply: S[g][l][d] += gamma * leaky_relu( sum_l w[g][l] * S[g][l][d] )   (dana.py:133-137)
// grid (D/64, G); 256 threads = 16 row groups x 16 float4 lanes: each row group sums every 16th row of its 64-channel
// slab, the 16 partials are added in group order through LDS (a fixed order: deterministic), then every thread adds the
// result back to its own rows. (Round 1 ran one thread per channel down all L rows: 48 workgroups, 84 us at L = 400.)
__global__ void __launch_bounds__(256)
ba_apply_kernel(float* __restrict__ S, const float* __restrict__ w, int L, int D, long ld, float gamma, float slope) {
  extern __shared__ float ws[];      // [L] weights | [16][64] partial sums
  float* part = ws + ((L + 3) & ~3);
  const int g = blockIdx.y;
  for (int l = threadIdx.x; l < L; l += blockDim.x) ws[l] = w[(long)g * L + l];
  __syncthreads();
  const int c4 = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const int d = blockIdx.x * 64 + c4 * 4;
  const bool ok = d + 3 < D;
  float* base = S + (long)g * L * ld + d;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (ok)
    for (int l = rg; l < L; l += 16) {
      const float4 v = *(const float4*)(base + (long)l * ld);
      const float wl = ws[l];
      acc.x += wl * v.x;
      acc.y += wl * v.y;
      acc.z += wl * v.z;
      acc.w += wl * v.w;
    }
  *(float4*)(part + rg * 64 + c4 * 4) = acc;
  __syncthreads();
  float4 tot = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float4 p = *(const float4*)(part + r * 64 + c4 * 4);
    tot.x += p.x;
    tot.y += p.y;
    tot.z += p.z;
    tot.w += p.w;
  }
  float4 add;
  add.x = gamma * (tot.x > 0.f ? tot.x : tot.x * slope);
  add.y = gamma * (tot.y > 0.f ? tot.y : tot.y * slope);
  add.z = gamma * (tot.z > 0.f ? tot.z : tot.z * slope);
  add.w = gamma * (tot.w > 0.f ? tot.w : tot.w * slope);
  if (ok)
    for (int l = rg; l < L; l += 16) {
      float4 v = *(const float4*)(base + (long)l * ld);
      v.x += add.x;
      v.y += add.y;
      v.z += add.z;
      v.w += add.w;
      *(float4*)(base + (long)l * ld) = v;
    }
}

// scores[r][seg*L + l] <- (softmax_l(scores[r][seg*L .. +L)) + ugamma * unary[b(r)][seg][l]) * out_scale;
// columns nseg*L .. ldp-1 (GEMM K padding) are zeroed. One wave per (row, segment-loop).
__global__ void __launch_bounds__(256)
attn_softmax_unary_kernel(float* __restrict__ scores, const float* __restrict__ unary, long rows, long rows_per_batch,
                          long unary_batch_stride, int nseg, int L, long ld, int kpad, float ugamma, float out_scale) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float* r = scores + row * ld;
  const float* u = unary + (row / rows_per_batch) * unary_batch_stride;
  constexpr int RV = 8;  // a segment of up to 512 scores stays in registers: one read and one write instead of 3 + 2
  if (L <= 64 * RV) {
    for (int sgm = 0; sgm < nseg; ++sgm) {
      float* x = r + sgm * L;
      float v[RV];
      float m = -FLT_MAX;
#pragma unroll
      for (int i = 0; i < RV; ++i) {
        const int l = lane + 64 * i;
        v[i] = l < L ? x[l] : -FLT_MAX;
        m = fmaxf(m, v[i]);
      }
      m = wave_max(m);
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < RV; ++i)
        if (lane + 64 * i < L) {  // (same per-lane summation order as the loop below)
          v[i] = expf(v[i] - m);
          s += v[i];
        }
      s = wave_sum(s);
#pragma unroll
      for (int i = 0; i < RV; ++i) {
        const int l = lane + 64 * i;
        if (l < L) x[l] = (v[i] / s + ugamma * u[sgm * L + l]) * out_scale;
      }
    }
    for (int l = nseg * L + lane; l < kpad; l += 64) r[l] = 0.f;
    return;
  }
  for (int sgm = 0; sgm < nseg; ++sgm) {
    float* x = r + sgm * L;
    float m = -FLT_MAX;
    for (int l = lane; l < L; l += 64) m = fmaxf(m, x[l]);
    m = wave_max(m);
    float s = 0.f;
    for (int l = lane; l < L; l += 64) {
      const float e = expf(x[l] - m);
      x[l] = e;
      s += e;
    }
    s = wave_sum(s);
    for (int l = lane; l < L; l += 64) x[l] = (x[l] / s + ugamma * u[sgm * L + l]) * out_scale;
  }
  for (int l = nseg * L + lane; l < kpad; l += 64) r[l] = 0.f;
}

}  // namespace

extern "C" {

int dana_rowdot(const float* x, const float* w, const float* bias, float* out, long rows, int dim, long ld,
                dana_stream_t stream) {
  DANA_CHECK_ARG(rows >= 0 && dim > 0 && dim % 4 == 0, "dana_rowdot: bad shape");
  if (rows == 0) return DANA_OK;
  DANA_CHECK_ARG(x && w && out, "dana_rowdot: null pointer");
  if (ld <= 0) ld = dim;
  DANA_CHECK_ARG(ld % 4 == 0, "dana_rowdot: ld %% 4 != 0");
  rowdot_kernel<<<dana_ceil_div(rows, 4), 256, 0, (hipStream_t)stream>>>(x, w, bias, out, rows, dim, ld);
  DANA_CHECK_LAUNCH("dana_rowdot");
  return DANA_OK;
}

int dana_softmax_rows(float* x, long groups, int length, long ld, dana_stream_t stream) {
  DANA_CHECK_ARG(groups >= 0 && length > 0, "dana_softmax_rows: bad shape");
  if (groups == 0) return DANA_OK;
  DANA_CHECK_ARG(x, "dana_softmax_rows: null pointer");
  if (ld <= 0) ld = length;
  softmax_rows_kernel<<<dana_ceil_div(groups, 4), 256, 0, (hipStream_t)stream>>>(x, groups, length, ld);
  DANA_CHECK_LAUNCH("dana_softmax_rows");
  return DANA_OK;
}

int dana_softmax_rows_to(const float* x, float* out, long groups, int length, long ld_in, long ld_out, dana_stream_t stream) {
  DANA_CHECK_ARG(groups >= 0 && length > 0, "dana_softmax_rows_to: bad shape");
  if (groups == 0) return DANA_OK;
  DANA_CHECK_ARG(x && out, "dana_softmax_rows_to: null pointer");
  if (ld_in <= 0) ld_in = length;
  if (ld_out <= 0) ld_out = length;
  softmax_rows_kernel<<<dana_ceil_div(groups, 4), 256, 0, (hipStream_t)stream>>>(out, groups, length, ld_out, x, ld_in);
  DANA_CHECK_LAUNCH("dana_softmax_rows_to");
  return DANA_OK;
}

int dana_ba_apply(float* s, const float* w, int groups, int length, int dim, long ld, float gamma, float slope,
                  dana_stream_t stream) {
  DANA_CHECK_ARG(groups >= 0 && length > 0 && dim > 0, "dana_ba_apply: bad shape");
  if (groups == 0) return DANA_OK;
  DANA_CHECK_ARG(s && w, "dana_ba_apply: null pointer");
  if (ld <= 0) ld = dim;
  DANA_CHECK_ARG(dim % 4 == 0 && ld % 4 == 0 && ((uintptr_t)s & 15) == 0, "dana_ba_apply: dim, ld must be multiples of 4 (float4 rows)");
  dim3 grid(dana_ceil_div(dim, 64), groups);
  const size_t lds = (size_t)(((length + 3) & ~3) + 16 * 64) * sizeof(float);
  ba_apply_kernel<<<grid, 256, lds, (hipStream_t)stream>>>(s, w, length, dim, ld, gamma, slope);
  DANA_CHECK_LAUNCH("dana_ba_apply");
  return DANA_OK;
}

int dana_attn_softmax_unary(float* scores, const float* unary, long rows, long rows_per_batch, long unary_batch_stride,
                            int nseg, int length, long ld, int kpad, float unary_gamma, float out_scale,
                            dana_stream_t stream) {
  DANA_CHECK_ARG(rows >= 0 && rows_per_batch > 0 && nseg > 0 && length > 0 && ld >= (long)nseg * length &&
                     kpad <= ld,
                 "dana_attn_softmax_unary: bad shape");
  if (rows == 0) return DANA_OK;
  DANA_CHECK_ARG(scores && unary, "dana_attn_softmax_unary: null pointer");
  attn_softmax_unary_kernel<<<dana_ceil_div(rows, 4), 256, 0, (hipStream_t)stream>>>(
      scores, unary, rows, rows_per_batch, unary_batch_stride > 0 ? unary_batch_stride : (long)nseg * length, nseg,
      length, ld, kpad, unary_gamma, out_scale);
  DANA_CHECK_LAUNCH("dana_attn_softmax_unary");
  return DANA_OK;
}

}  // extern "C"
