// Winograd F(2x2, 3x3) for the stride-1 3x3 convolutions with many channels (layer3 / layer4 conv2 of
// the Bottlenecks, resnet.py:73-74, and RPN_Conv, rpn.py:28): 2.25x fewer multiplies than the direct
// implicit GEMM, still exact-fp32 MFMA arithmetic (the transforms only add / halve).
//
//   Y = A^T [ (G g G^T) .* (B^T d B) ] A        d = 4x4 input patch, g = 3x3 filter, Y = 2x2 outputs
//
//   wino_filter_kernel   U[xi][cout][cin]  = G g G^T            once per weight version
//   wino_input_kernel    V[xi][tile][cin]  = B^T d B            HBM-bound, float4 lanes over channels,
//                                                               branch-free buffer loads (padding -> 0)
//   dana_gemm_nt(batch = 16)  M[xi][tile][cout] = V[xi] . U[xi]^T   the MFMA kernel of igemm.hip
//   wino_output_kernel   out = relu?( (A^T M A) * scale + shift )   HBM-bound, writes NHWC with a row stride
//
// Planes are [16][tiles][C]: every plane is a plain K-contiguous GEMM operand, so the 16 products are
// ONE batched launch with 16x the tiles of a single GEMM (good for the 256 CUs).
#include "common.h"
#include "../../include/dana_hip.h"

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned OOB = 0x80000000u;

__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }

// packed weight [cout][3][3][cin] -> U[16][cout][cin]
__global__ void __launch_bounds__(256)
wino_filter_kernel(const float* __restrict__ w, float* __restrict__ U, int cout, int cin) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)cout * cin) return;
  const int ci = (int)(i % cin);
  const long co = i / cin;
  float g[3][3];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) g[r][c] = w[((co * 3 + r) * 3 + c) * cin + ci];
  float t[4][3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    t[0][c] = g[0][c];
    t[1][c] = 0.5f * (g[0][c] + g[1][c] + g[2][c]);
    t[2][c] = 0.5f * (g[0][c] - g[1][c] + g[2][c]);
    t[3][c] = g[2][c];
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float u0 = t[r][0], u1 = 0.5f * (t[r][0] + t[r][1] + t[r][2]), u2 = 0.5f * (t[r][0] - t[r][1] + t[r][2]),
                u3 = t[r][2];
    const long plane = (long)cout * cin;
    U[(r * 4 + 0) * plane + i] = u0;
    U[(r * 4 + 1) * plane + i] = u1;
    U[(r * 4 + 2) * plane + i] = u2;
    U[(r * 4 + 3) * plane + i] = u3;
  }
}

// one lane = (tile, 4 channels)
__global__ void __launch_bounds__(256)
wino_input_kernel(const float* __restrict__ in, float* __restrict__ V, int H, int W, int C4, int th, int tw,
                  long tiles, int lda, unsigned in_bytes) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= tiles * C4) return;
  const int c4 = (int)(idx % C4);
  const long t = idx / C4;
  const int j = (int)(t % tw);
  const int i = (int)((t / tw) % th);
  const int img = (int)(t / ((long)tw * th));
  const __amdgpu_buffer_rsrc_t src = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, (int)in_bytes, 0x00020000);
  float4 d[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int y = 2 * i - 1 + r;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int x = 2 * j - 1 + c;
      const bool ok = y >= 0 && y < H && x >= 0 && x < W;
      const unsigned off = ok ? (unsigned)((((img * H + y) * W + x) * lda + c4 * 4) * 4) : OOB;
      u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(src, (int)off, 0, 0);
      d[r][c] = *(float4*)&v;
    }
  }
  float4 tm[4][4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {  // B^T d
    tm[0][c] = f4sub(d[0][c], d[2][c]);
    tm[1][c] = f4add(d[1][c], d[2][c]);
    tm[2][c] = f4sub(d[2][c], d[1][c]);
    tm[3][c] = f4sub(d[1][c], d[3][c]);
  }
  const long plane = tiles * (long)C4;  // in float4 units
  float4* out = (float4*)V + t * C4 + c4;
#pragma unroll
  for (int r = 0; r < 4; ++r) {  // (.) B
    out[(r * 4 + 0) * plane] = f4sub(tm[r][0], tm[r][2]);
    out[(r * 4 + 1) * plane] = f4add(tm[r][1], tm[r][2]);
    out[(r * 4 + 2) * plane] = f4sub(tm[r][2], tm[r][1]);
    out[(r * 4 + 3) * plane] = f4sub(tm[r][1], tm[r][3]);
  }
}

// one lane = (tile, 4 output channels)
__global__ void __launch_bounds__(256)
wino_output_kernel(const float* __restrict__ M, float* __restrict__ out, const float* __restrict__ scale,
                   const float* __restrict__ shift, const float* __restrict__ mask, long ldm, int H, int W, int N4,
                   int th, int tw, long tiles, long ldc, int relu) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= tiles * N4) return;
  const int n4 = (int)(idx % N4);
  const long t = idx / N4;
  const int j = (int)(t % tw);
  const int i = (int)((t / tw) % th);
  const long img = t / ((long)tw * th);
  const long plane = tiles * (long)N4;
  const float4* mp = (const float4*)M + t * N4 + n4;
  float4 m[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) m[r][c] = mp[(r * 4 + c) * plane];
  float4 s[2][4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {  // A^T m
    s[0][c] = f4add(f4add(m[0][c], m[1][c]), m[2][c]);
    s[1][c] = f4sub(f4sub(m[1][c], m[2][c]), m[3][c]);
  }
  const float4 sc = scale ? ((const float4*)scale)[n4] : make_float4(1, 1, 1, 1);
  const float4 sh = shift ? ((const float4*)shift)[n4] : make_float4(0, 0, 0, 0);
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int y = 2 * i + r;
    if (y >= H) break;
    float4 y0 = f4add(f4add(s[r][0], s[r][1]), s[r][2]);
    float4 y1 = f4sub(f4sub(s[r][1], s[r][2]), s[r][3]);
    float4 o[2] = {y0, y1};
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int x = 2 * j + c;
      if (x >= W) break;
      float4 v = make_float4(o[c].x * sc.x + sh.x, o[c].y * sc.y + sh.y, o[c].z * sc.z + sh.z, o[c].w * sc.w + sh.w);
      if (relu) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
      if (mask) {  // ReLU adjoint (data-gradient use): zero where the masking activation is <= 0
        const float4 k = *(const float4*)(mask + ((img * H + y) * W + x) * ldm + n4 * 4);
        v = make_float4(k.x > 0.f ? v.x : 0.f, k.y > 0.f ? v.y : 0.f, k.z > 0.f ? v.z : 0.f, k.w > 0.f ? v.w : 0.f);
      }
      *(float4*)(out + ((img * H + y) * W + x) * ldc + n4 * 4) = v;
    }
  }
}

// ---- F(4x4, 3x3): 36 planes, 4x fewer multiplies than the direct conv (F(2x2,3x3): 2.25x), and the transformed
// tensors are SMALLER than F(2x2)'s (36/16 of the pixels instead of 16/4). The transform constants (up to 8) cost
// about one decimal digit: ~1e-5 relative error instead of ~1e-6, still inside the 1e-4 score tolerance.
//   B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
//   G   = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1]
//   A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
__device__ __forceinline__ float4 f4s(float4 a, float s) { return make_float4(a.x * s, a.y * s, a.z * s, a.w * s); }
__device__ __forceinline__ float4 f4ma(float4 a, float s, float4 b) {  // a * s + b
  return make_float4(a.x * s + b.x, a.y * s + b.y, a.z * s + b.z, a.w * s + b.w);
}

// rows of B^T applied to a 6-vector of float4
__device__ __forceinline__ void bt6(const float4 (&d)[6], float4 (&o)[6]) {
  o[0] = f4add(f4ma(d[2], -5.f, f4s(d[0], 4.f)), d[4]);
  o[1] = f4add(f4ma(f4add(d[1], d[2]), -4.f, d[3]), d[4]);
  o[2] = f4add(f4ma(f4sub(d[1], d[2]), 4.f, f4sub(d[4], d[3])), make_float4(0.f, 0.f, 0.f, 0.f));
  o[3] = f4add(f4ma(f4sub(d[3], d[1]), 2.f, f4sub(d[4], d[2])), make_float4(0.f, 0.f, 0.f, 0.f));
  o[4] = f4add(f4ma(f4sub(d[1], d[3]), 2.f, f4sub(d[4], d[2])), make_float4(0.f, 0.f, 0.f, 0.f));
  o[5] = f4add(f4ma(d[3], -5.f, f4s(d[1], 4.f)), d[5]);
}

__global__ void __launch_bounds__(256)
wino4_filter_kernel(const float* __restrict__ w, float* __restrict__ U, int cout, int cin) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)cout * cin) return;
  const int ci = (int)(i % cin);
  const long co = i / cin;
  float g[3][3];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) g[r][c] = w[((co * 3 + r) * 3 + c) * cin + ci];
  float t[6][3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {  // G g
    const float a = g[0][c], b = g[1][c], d = g[2][c];
    t[0][c] = 0.25f * a;
    t[1][c] = -(a + b + d) * (1.f / 6.f);
    t[2][c] = (-a + b - d) * (1.f / 6.f);
    t[3][c] = a * (1.f / 24.f) + b * (1.f / 12.f) + d * (1.f / 6.f);
    t[4][c] = a * (1.f / 24.f) - b * (1.f / 12.f) + d * (1.f / 6.f);
    t[5][c] = d;
  }
  const long plane = (long)cout * cin;
#pragma unroll
  for (int r = 0; r < 6; ++r) {  // (.) G^T
    const float a = t[r][0], b = t[r][1], d = t[r][2];
    U[(r * 6 + 0) * plane + i] = 0.25f * a;
    U[(r * 6 + 1) * plane + i] = -(a + b + d) * (1.f / 6.f);
    U[(r * 6 + 2) * plane + i] = (-a + b - d) * (1.f / 6.f);
    U[(r * 6 + 3) * plane + i] = a * (1.f / 24.f) + b * (1.f / 12.f) + d * (1.f / 6.f);
    U[(r * 6 + 4) * plane + i] = a * (1.f / 24.f) - b * (1.f / 12.f) + d * (1.f / 6.f);
    U[(r * 6 + 5) * plane + i] = d;
  }
}

// one lane = (tile of 4x4 outputs, 4 channels): V[36][tiles][C]
__global__ void __launch_bounds__(256)
wino4_input_kernel(const float* __restrict__ in, float* __restrict__ V, int H, int W, int C4, int th, int tw,
                   long tiles, int lda, unsigned in_bytes, long plane_tiles = 0, long tile_off = 0) {
  // plane_tiles / tile_off: this launch fills tiles [tile_off, tile_off + tiles) of planes that hold plane_tiles tiles
  // (two image groups -- query and support batch -- share one batched plane GEMM); 0 = the planes are this launch's own
  if (plane_tiles == 0) plane_tiles = tiles;
  // 32-bit index math (tiles * C4 < 2^31 is checked by the host): the 64-bit divides cost more than the transform
  const unsigned idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (unsigned)(tiles * C4)) return;
  const unsigned t = idx / (unsigned)C4;
  const int c4 = (int)(idx - t * (unsigned)C4);
  const unsigned tr = t / (unsigned)tw;
  const int j = (int)(t - tr * (unsigned)tw);
  const int img = (int)(tr / (unsigned)th);
  const int i = (int)(tr - (unsigned)img * (unsigned)th);
  const __amdgpu_buffer_rsrc_t src = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, (int)in_bytes, 0x00020000);
  float4 tm[6][6];  // B^T d, built column by column (6 loads in flight per column)
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    const int x = 4 * j - 1 + c;
    float4 col[6], o[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const int y = 4 * i - 1 + r;
      const bool ok = y >= 0 && y < H && x >= 0 && x < W;
      const unsigned off = ok ? (unsigned)((((img * H + y) * W + x) * lda + c4 * 4) * 4) : OOB;
      u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(src, (int)off, 0, 0);
      col[r] = *(float4*)&v;
    }
    bt6(col, o);
#pragma unroll
    for (int r = 0; r < 6; ++r) tm[r][c] = o[r];
  }
  const long plane = plane_tiles * (long)C4;  // in float4 units
  float4* out = (float4*)V + (tile_off + t) * C4 + c4;
#pragma unroll
  for (int r = 0; r < 6; ++r) {  // (.) B  ==  rows of B^T applied to the row vector
    float4 o[6];
    bt6(tm[r], o);
#pragma unroll
    for (int c = 0; c < 6; ++c) out[(r * 6 + c) * plane] = o[c];
  }
}

// rows of A^T applied to a 6-vector -> 4-vector
__device__ __forceinline__ void at6(const float4 (&m)[6], float4 (&o)[4]) {
  const float4 p = f4add(m[1], m[2]), q = f4sub(m[1], m[2]), r = f4add(m[3], m[4]), s = f4sub(m[3], m[4]);
  o[0] = f4add(f4add(m[0], p), r);
  o[1] = f4ma(s, 2.f, q);
  o[2] = f4ma(r, 4.f, p);
  o[3] = f4add(f4ma(s, 8.f, q), m[5]);
}

// one lane = (tile, 4 output channels)
__global__ void __launch_bounds__(256)
wino4_output_kernel(const float* __restrict__ M, float* __restrict__ out, const float* __restrict__ scale,
                    const float* __restrict__ shift, const float* __restrict__ mask, long ldm, int H, int W, int N4,
                    int th, int tw, long tiles, long ldc, int relu, long plane_tiles = 0, long tile_off = 0) {
  const unsigned idx = blockIdx.x * blockDim.x + threadIdx.x;  // (32-bit index math, see wino4_input_kernel)
  if (idx >= (unsigned)(tiles * N4)) return;
  if (plane_tiles == 0) plane_tiles = tiles;
  const unsigned t = idx / (unsigned)N4;
  const int n4 = (int)(idx - t * (unsigned)N4);
  const unsigned tr = t / (unsigned)tw;
  const int j = (int)(t - tr * (unsigned)tw);
  const long img = (long)(tr / (unsigned)th);
  const int i = (int)(tr - (unsigned)img * (unsigned)th);
  const long plane = plane_tiles * (long)N4;
  const float4* mp = (const float4*)M + tile_off * N4 + (long)idx;
  float4 s[4][6];  // A^T m, column by column
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    float4 col[6], o[4];
#pragma unroll
    for (int r = 0; r < 6; ++r) col[r] = mp[(r * 6 + c) * plane];
    at6(col, o);
#pragma unroll
    for (int r = 0; r < 4; ++r) s[r][c] = o[r];
  }
  const float4 sc = scale ? ((const float4*)scale)[n4] : make_float4(1, 1, 1, 1);
  const float4 sh = shift ? ((const float4*)shift)[n4] : make_float4(0, 0, 0, 0);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int y = 4 * i + r;
    if (y >= H) break;
    float4 o[4];
    at6(s[r], o);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int x = 4 * j + c;
      if (x >= W) break;
      float4 v = make_float4(o[c].x * sc.x + sh.x, o[c].y * sc.y + sh.y, o[c].z * sc.z + sh.z, o[c].w * sc.w + sh.w);
      if (relu) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
      if (mask) {
        const float4 k = *(const float4*)(mask + ((img * H + y) * W + x) * ldm + n4 * 4);
        v = make_float4(k.x > 0.f ? v.x : 0.f, k.y > 0.f ? v.y : 0.f, k.z > 0.f ? v.z : 0.f, k.w > 0.f ? v.w : 0.f);
      }
      *(float4*)(out + ((img * H + y) * W + x) * ldc + n4 * 4) = v;
    }
  }
}

// ---- weight gradient in the F(4x4,3x3) domain: dU[xi] = sum_tiles dM[xi]^T V[xi], dM = A dY A^T, dW = G^T dU G ----
// rows of A (6x4) applied to a 4-vector -> 6-vector: A = [1 0 0 0; 1 1 1 1; 1 -1 1 -1; 1 2 4 8; 1 -2 4 -8; 0 0 0 1]
__device__ __forceinline__ void a4to6(const float4 (&y)[4], float4 (&o)[6]) {
  const float4 p = f4add(y[0], y[2]), q = f4add(y[1], y[3]);
  const float4 r = f4ma(y[2], 4.f, y[0]), t = f4ma(y[3], 8.f, f4s(y[1], 2.f));
  o[0] = y[0];
  o[1] = f4add(p, q);
  o[2] = f4sub(p, q);
  o[3] = f4add(r, t);
  o[4] = f4sub(r, t);
  o[5] = y[3];
}

// one lane = (tile, 4 output channels): dM[36][tiles][N] from dY [B*H*W][ldy]; outputs outside the image are zeros
__global__ void __launch_bounds__(256)
wino4_outgrad_kernel(const float* __restrict__ dy, float* __restrict__ dM, int H, int W, int N4, int th, int tw,
                     long tiles, long ldy) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= tiles * N4) return;
  const int n4 = (int)(idx % N4);
  const long t = idx / N4;
  const int j = (int)(t % tw);
  const int i = (int)((t / tw) % th);
  const long img = t / ((long)tw * th);
  float4 tm[6][4];  // A dy, column by column
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int x = 4 * j + c;
    float4 col[4], o[6];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int y = 4 * i + r;
      col[r] = (y < H && x < W) ? *(const float4*)(dy + ((img * H + y) * W + x) * ldy + n4 * 4)
                                : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    a4to6(col, o);
#pragma unroll
    for (int r = 0; r < 6; ++r) tm[r][c] = o[r];
  }
  const long plane = tiles * (long)N4;
  float4* out = (float4*)dM + t * N4 + n4;
#pragma unroll
  for (int r = 0; r < 6; ++r) {  // (.) A^T
    float4 o[6];
    a4to6(tm[r], o);
#pragma unroll
    for (int c = 0; c < 6; ++c) out[(r * 6 + c) * plane] = o[c];
  }
}

// dW[co][3][3][ci] (+)= scale[co] * (G^T dU G): one lane = (co, ci)
__global__ void __launch_bounds__(256)
wino4_filtergrad_kernel(const float* __restrict__ dU, float* __restrict__ dW, const float* __restrict__ row_scale,
                        int cout, int cin, int accumulate) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)cout * cin) return;
  const int ci = (int)(i % cin);
  const long co = i / cin;
  const long plane = (long)cout * cin;
  float t[3][6];  // G^T dU  (G^T = [1/4 -1/6 -1/6 1/24 1/24 0; 0 -1/6 1/6 1/12 -1/12 0; 0 -1/6 -1/6 1/6 1/6 1])
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    float u[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) u[r] = dU[(r * 6 + c) * plane + i];
    t[0][c] = 0.25f * u[0] - (u[1] + u[2]) * (1.f / 6.f) + (u[3] + u[4]) * (1.f / 24.f);
    t[1][c] = (u[2] - u[1]) * (1.f / 6.f) + (u[3] - u[4]) * (1.f / 12.f);
    t[2][c] = -(u[1] + u[2]) * (1.f / 6.f) + (u[3] + u[4]) * (1.f / 6.f) + u[5];
  }
  const float sc = row_scale ? row_scale[co] : 1.f;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const float* u = t[r];
    const float g0 = 0.25f * u[0] - (u[1] + u[2]) * (1.f / 6.f) + (u[3] + u[4]) * (1.f / 24.f);
    const float g1 = (u[2] - u[1]) * (1.f / 6.f) + (u[3] - u[4]) * (1.f / 12.f);
    const float g2 = -(u[1] + u[2]) * (1.f / 6.f) + (u[3] + u[4]) * (1.f / 6.f) + u[5];
    float* o = dW + ((co * 3 + r) * 3) * cin + ci;
    o[0] = (accumulate ? o[0] : 0.f) + sc * g0;
    o[cin] = (accumulate ? o[cin] : 0.f) + sc * g1;
    o[2 * cin] = (accumulate ? o[2 * cin] : 0.f) + sc * g2;
  }
}

struct WinoPlan {
  int th, tw;
  long tiles;
  size_t v_bytes, m_bytes, total;
};
WinoPlan wino_plan(int batch, int h, int w, int cin, int cout, int m = 2) {  // m = output tile edge: 2 or 4
  WinoPlan p;
  const int planes = (m + 2) * (m + 2);
  p.th = (h + m - 1) / m;
  p.tw = (w + m - 1) / m;
  p.tiles = (long)batch * p.th * p.tw;
  p.v_bytes = dana_align_up((size_t)planes * p.tiles * cin * 4, 256);
  p.m_bytes = dana_align_up((size_t)planes * p.tiles * cout * 4, 256);
  p.total = p.v_bytes + p.m_bytes;
  return p;
}

}  // namespace

extern "C" {

int dana_winograd_filter_transform(const float* w_packed, float* u, int cout, int cin, dana_stream_t stream) {
  DANA_CHECK_ARG(w_packed && u && cout > 0 && cin > 0, "dana_winograd_filter_transform: bad args");
  const long total = (long)cout * cin;
  wino_filter_kernel<<<dana_ceil_div(total, 256), 256, 0, (hipStream_t)stream>>>(w_packed, u, cout, cin);
  DANA_CHECK_LAUNCH("dana_winograd_filter_transform");
  return DANA_OK;
}

size_t dana_conv3x3_winograd_workspace_bytes(int batch, int h, int w, int cin, int cout) {
  if (batch <= 0 || h <= 0 || w <= 0 || cin <= 0 || cout <= 0) return 0;
  return wino_plan(batch, h, w, cin, cout).total;
}

int dana_conv3x3_winograd_nhwc(const float* input, const float* u, float* output, const float* scale,
                               const float* shift, int batch, int h, int w, int cin, int cout, long in_pix_stride,
                               long out_pix_stride, int flags, void* workspace, size_t workspace_bytes,
                               dana_stream_t stream) {
  return dana_conv3x3_winograd_nhwc_masked(input, u, output, scale, shift, nullptr, batch, h, w, cin, cout, in_pix_stride,
                                           out_pix_stride, 0, flags, workspace, workspace_bytes, stream);
}

int dana_conv3x3_winograd_nhwc_masked(const float* input, const float* u, float* output, const float* scale,
                                      const float* shift, const float* mask_act, int batch, int h, int w, int cin,
                                      int cout, long in_pix_stride, long out_pix_stride, long mask_pix_stride,
                                      int flags, void* workspace, size_t workspace_bytes, dana_stream_t stream) {
  DANA_CHECK_ARG(batch >= 0 && h > 0 && w > 0 && cin > 0 && cout > 0 && cin % 4 == 0 && cout % 4 == 0,
                 "dana_conv3x3_winograd_nhwc: bad shape");
  if (batch == 0) return DANA_OK;
  DANA_CHECK_ARG(input && u && output, "dana_conv3x3_winograd_nhwc: null pointer");
  const long lda = in_pix_stride > 0 ? in_pix_stride : cin;
  const long ldc = out_pix_stride > 0 ? out_pix_stride : cout;
  const long ldm = mask_pix_stride > 0 ? mask_pix_stride : cout;
  DANA_CHECK_ARG(!mask_act || (ldm % 4 == 0 && ((uintptr_t)mask_act & 15) == 0),
                 "dana_conv3x3_winograd_nhwc: mask rows must be 16-byte aligned");
  DANA_CHECK_ARG(lda % 4 == 0 && ldc % 4 == 0 && ((uintptr_t)input & 15) == 0 && ((uintptr_t)output & 15) == 0,
                 "dana_conv3x3_winograd_nhwc: strides / pointers must be 16-byte aligned");
  const long in_bytes = (long)batch * h * w * lda * 4;
  DANA_CHECK_ARG(in_bytes < (long)OOB, "dana_conv3x3_winograd_nhwc: input span >= 2 GiB; split the batch");
  const WinoPlan p = wino_plan(batch, h, w, cin, cout);
  if (!workspace || workspace_bytes < p.total) {
    dana_set_error("dana_conv3x3_winograd_nhwc: workspace %zu < %zu", workspace_bytes, p.total);
    return DANA_ERR_WORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  float* V = (float*)workspace;
  float* M = (float*)((char*)workspace + p.v_bytes);
  const int C4 = cin / 4, N4 = cout / 4;
  wino_input_kernel<<<dana_ceil_div(p.tiles * C4, 256), 256, 0, s>>>(input, V, h, w, C4, p.th, p.tw, p.tiles, (int)lda,
                                                                    (unsigned)in_bytes);
  DANA_CHECK_LAUNCH("dana_conv3x3_winograd_nhwc(input transform)");
  int rc = dana_gemm_nt(V, u, M, nullptr, nullptr, nullptr, (int)p.tiles, cout, cin, cin, cin, cout, 0, 16,
                        p.tiles * cin, (long)cout * cin, p.tiles * cout, 1.f, 0, stream);
  if (rc) return rc;
  wino_output_kernel<<<dana_ceil_div(p.tiles * N4, 256), 256, 0, s>>>(M, output, scale, shift, mask_act, ldm, h, w, N4,
                                                                     p.th, p.tw, p.tiles, ldc,
                                                                     (flags & DANA_EPI_RELU) ? 1 : 0);
  DANA_CHECK_LAUNCH("dana_conv3x3_winograd_nhwc(output transform)");
  return DANA_OK;
}

size_t dana_conv3x3_wgrad_winograd4_workspace_bytes(int batch, int h, int w, int cin, int cout) {
  if (batch <= 0 || h <= 0 || w <= 0 || cin <= 0 || cout <= 0) return 0;
  const WinoPlan p = wino_plan(batch, h, w, cin, cout, 4);
  const size_t du = dana_align_up((size_t)36 * cout * cin * 4, 256);
  return p.total + du + dana_align_up(dana_wgrad_tn_batched_workspace(36, (int)p.tiles, cout, cin), 256);
}

static int wgrad_winograd4_impl(const float* grad_out, const float* input, const float* v_saved, float* grad_weight,
                                int batch, int h, int w, int cin, int cout, long in_pix_stride, long grad_pix_stride,
                                const float* row_scale, int accumulate, void* workspace, size_t workspace_bytes,
                                dana_stream_t stream);

int dana_conv3x3_wgrad_winograd4(const float* grad_out, const float* input, float* grad_weight, int batch, int h, int w,
                                 int cin, int cout, long in_pix_stride, long grad_pix_stride, const float* row_scale,
                                 int accumulate, void* workspace, size_t workspace_bytes, dana_stream_t stream) {
  DANA_CHECK_ARG(input || batch == 0, "dana_conv3x3_wgrad_winograd4: null pointer");
  return wgrad_winograd4_impl(grad_out, input, nullptr, grad_weight, batch, h, w, cin, cout, in_pix_stride, grad_pix_stride,
                              row_scale, accumulate, workspace, workspace_bytes, stream);
}

/* the same with the input's transform handed in: v = the V planes [36][tiles][cin] the forward's
 * dana_conv3x3_winograd4_nhwc left at the start of ITS workspace (kept by the caller) -- no second input transform */
int dana_conv3x3_wgrad_winograd4_v(const float* grad_out, const float* v, float* grad_weight, int batch, int h, int w,
                                   int cin, int cout, long grad_pix_stride, const float* row_scale, int accumulate,
                                   void* workspace, size_t workspace_bytes, dana_stream_t stream) {
  DANA_CHECK_ARG(v || batch == 0, "dana_conv3x3_wgrad_winograd4_v: null pointer");
  DANA_CHECK_ARG(((uintptr_t)v & 15) == 0, "dana_conv3x3_wgrad_winograd4_v: v must be 16-byte aligned");
  return wgrad_winograd4_impl(grad_out, nullptr, v, grad_weight, batch, h, w, cin, cout, 0, grad_pix_stride, row_scale,
                              accumulate, workspace, workspace_bytes, stream);
}

static int wgrad_winograd4_impl(const float* grad_out, const float* input, const float* v_saved, float* grad_weight,
                                int batch, int h, int w, int cin, int cout, long in_pix_stride, long grad_pix_stride,
                                const float* row_scale, int accumulate, void* workspace, size_t workspace_bytes,
                                dana_stream_t stream) {
  DANA_CHECK_ARG(batch >= 0 && h > 0 && w > 0 && cin > 0 && cout > 0 && cin % 64 == 0 && cout % 4 == 0,
                 "dana_conv3x3_wgrad_winograd4: bad shape (cin %% 64, cout %% 4)");
  if (batch == 0) return DANA_OK;
  DANA_CHECK_ARG(grad_out && (input || v_saved) && grad_weight, "dana_conv3x3_wgrad_winograd4: null pointer");
  const long lda = in_pix_stride > 0 ? in_pix_stride : cin;
  const long ldy = grad_pix_stride > 0 ? grad_pix_stride : cout;
  DANA_CHECK_ARG(lda % 4 == 0 && ldy % 4 == 0 && ((uintptr_t)input & 15) == 0 && ((uintptr_t)grad_out & 15) == 0,
                 "dana_conv3x3_wgrad_winograd4: strides / pointers must be 16-byte aligned");
  const long in_bytes = (long)batch * h * w * lda * 4;
  DANA_CHECK_ARG(in_bytes < (long)OOB, "dana_conv3x3_wgrad_winograd4: input span >= 2 GiB; split the batch");
  const size_t need = dana_conv3x3_wgrad_winograd4_workspace_bytes(batch, h, w, cin, cout);
  if (!workspace || workspace_bytes < need) {
    dana_set_error("dana_conv3x3_wgrad_winograd4: workspace %zu < %zu", workspace_bytes, need);
    return DANA_ERR_WORKSPACE;
  }
  const WinoPlan p = wino_plan(batch, h, w, cin, cout, 4);
  hipStream_t s = (hipStream_t)stream;
  const float* V = v_saved ? v_saved : (const float*)workspace;  // [36][tiles][cin]
  float* dM = (float*)((char*)workspace + p.v_bytes);      // [36][tiles][cout]
  float* dU = (float*)((char*)workspace + p.total);        // [36][cout][cin]
  const size_t du = dana_align_up((size_t)36 * cout * cin * 4, 256);
  void* gws = (char*)workspace + p.total + du;
  const int C4 = cin / 4, N4 = cout / 4;
  DANA_CHECK_ARG(p.tiles * (long)(C4 > N4 ? C4 : N4) < (1L << 31), "dana_conv3x3_wgrad_winograd4: too many tiles x channels");
  if (!v_saved) {
    wino4_input_kernel<<<dana_ceil_div(p.tiles * C4, 256), 256, 0, s>>>(input, (float*)workspace, h, w, C4, p.th, p.tw, p.tiles,
                                                                       (int)lda, (unsigned)in_bytes);
    DANA_CHECK_LAUNCH("dana_conv3x3_wgrad_winograd4(input transform)");
  }
  wino4_outgrad_kernel<<<dana_ceil_div(p.tiles * N4, 256), 256, 0, s>>>(grad_out, dM, h, w, N4, p.th, p.tw, p.tiles, ldy);
  DANA_CHECK_LAUNCH("dana_conv3x3_wgrad_winograd4(output-gradient transform)");
  int rc = dana_wgrad_tn_batched(dM, V, dU, 36, (int)p.tiles, cout, cin, p.tiles * cout, p.tiles * cin, gws,
                                 workspace_bytes - (p.total + du), (void*)s);
  if (rc) return rc;
  const long total = (long)cout * cin;
  wino4_filtergrad_kernel<<<dana_ceil_div(total, 256), 256, 0, s>>>(dU, grad_weight, row_scale, cout, cin, accumulate);
  DANA_CHECK_LAUNCH("dana_conv3x3_wgrad_winograd4(filter-gradient transform)");
  return DANA_OK;
}

int dana_winograd4_filter_transform(const float* w_packed, float* u, int cout, int cin, dana_stream_t stream) {
  DANA_CHECK_ARG(w_packed && u && cout > 0 && cin > 0, "dana_winograd4_filter_transform: bad args");
  const long total = (long)cout * cin;
  wino4_filter_kernel<<<dana_ceil_div(total, 256), 256, 0, (hipStream_t)stream>>>(w_packed, u, cout, cin);
  DANA_CHECK_LAUNCH("dana_winograd4_filter_transform");
  return DANA_OK;
}

size_t dana_conv3x3_winograd4_workspace_bytes(int batch, int h, int w, int cin, int cout) {
  if (batch <= 0 || h <= 0 || w <= 0 || cin <= 0 || cout <= 0) return 0;
  return wino_plan(batch, h, w, cin, cout, 4).total;
}

int dana_conv3x3_winograd4_nhwc_masked(const float* input, const float* u, float* output, const float* scale,
                                       const float* shift, const float* mask_act, int batch, int h, int w, int cin,
                                       int cout, long in_pix_stride, long out_pix_stride, long mask_pix_stride,
                                       int flags, void* workspace, size_t workspace_bytes, dana_stream_t stream) {
  DANA_CHECK_ARG(batch >= 0 && h > 0 && w > 0 && cin > 0 && cout > 0 && cin % 4 == 0 && cout % 4 == 0,
                 "dana_conv3x3_winograd4_nhwc: bad shape");
  if (batch == 0) return DANA_OK;
  DANA_CHECK_ARG(input && u && output, "dana_conv3x3_winograd4_nhwc: null pointer");
  const long lda = in_pix_stride > 0 ? in_pix_stride : cin;
  const long ldc = out_pix_stride > 0 ? out_pix_stride : cout;
  const long ldm = mask_pix_stride > 0 ? mask_pix_stride : cout;
  DANA_CHECK_ARG(!mask_act || (ldm % 4 == 0 && ((uintptr_t)mask_act & 15) == 0),
                 "dana_conv3x3_winograd4_nhwc: mask rows must be 16-byte aligned");
  DANA_CHECK_ARG(lda % 4 == 0 && ldc % 4 == 0 && ((uintptr_t)input & 15) == 0 && ((uintptr_t)output & 15) == 0,
                 "dana_conv3x3_winograd4_nhwc: strides / pointers must be 16-byte aligned");
  const long in_bytes = (long)batch * h * w * lda * 4;
  DANA_CHECK_ARG(in_bytes < (long)OOB, "dana_conv3x3_winograd4_nhwc: input span >= 2 GiB; split the batch");
  const WinoPlan p = wino_plan(batch, h, w, cin, cout, 4);
  if (!workspace || workspace_bytes < p.total) {
    dana_set_error("dana_conv3x3_winograd4_nhwc: workspace %zu < %zu", workspace_bytes, p.total);
    return DANA_ERR_WORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  float* V = (float*)workspace;
  float* M = (float*)((char*)workspace + p.v_bytes);
  const int C4 = cin / 4, N4 = cout / 4;
  DANA_CHECK_ARG(p.tiles * (long)(C4 > N4 ? C4 : N4) < (1L << 31), "dana_conv3x3_winograd4_nhwc: too many tiles x channels");
  wino4_input_kernel<<<dana_ceil_div(p.tiles * C4, 256), 256, 0, s>>>(input, V, h, w, C4, p.th, p.tw, p.tiles, (int)lda,
                                                                     (unsigned)in_bytes);
  DANA_CHECK_LAUNCH("dana_conv3x3_winograd4_nhwc(input transform)");
  // (DANA_W_SPLIT3: u = dana_split_weight of the 36 transformed filter planes, [36][3][cout][kp] bf16)
  const bool w3 = (flags & DANA_W_SPLIT3) != 0;
  const long kp = (cin + 15) / 16 * 16;
  int rc = dana_gemm_nt(V, u, M, nullptr, nullptr, nullptr, (int)p.tiles, cout, cin, cin, w3 ? kp : cin, cout, 0, 36,
                        p.tiles * cin, w3 ? 3 * cout * kp : (long)cout * cin, p.tiles * cout, 1.f, w3 ? DANA_W_SPLIT3 : 0,
                        stream);
  if (rc) return rc;
  wino4_output_kernel<<<dana_ceil_div(p.tiles * N4, 256), 256, 0, s>>>(M, output, scale, shift, mask_act, ldm, h, w, N4,
                                                                      p.th, p.tw, p.tiles, ldc,
                                                                      (flags & DANA_EPI_RELU) ? 1 : 0);
  DANA_CHECK_LAUNCH("dana_conv3x3_winograd4_nhwc(output transform)");
  return DANA_OK;
}

size_t dana_conv3x3_winograd4_dual_workspace_bytes(int n0, int h0, int w0, int n1, int h1, int w1, int cin, int cout) {
  if (n0 < 0 || n1 < 0 || cin <= 0 || cout <= 0) return 0;
  const long t0 = n0 > 0 ? wino_plan(n0, h0, w0, cin, cout, 4).tiles : 0, t1 = n1 > 0 ? wino_plan(n1, h1, w1, cin, cout, 4).tiles : 0;
  return dana_align_up((size_t)36 * (t0 + t1) * cin * 4, 256) + dana_align_up((size_t)36 * (t0 + t1) * cout * 4, 256);
}

int dana_conv3x3_winograd4_nhwc_dual(const float* input, const float* u, float* out0, float* out1, const float* scale,
                                     const float* shift, int n0, int h0, int w0, int n1, int h1, int w1, int cin,
                                     int cout, long in_pix_stride, long out0_pix_stride, long out1_pix_stride, int flags,
                                     void* workspace, size_t workspace_bytes, dana_stream_t stream) {
  return dana_conv3x3_winograd4_nhwc_dual_masked(input, u, out0, out1, scale, shift, nullptr, nullptr, n0, h0, w0, n1, h1,
                                                 w1, cin, cout, in_pix_stride, out0_pix_stride, out1_pix_stride, 0, 0,
                                                 flags, workspace, workspace_bytes, stream);
}

int dana_conv3x3_winograd4_nhwc_dual_masked(const float* input, const float* u, float* out0, float* out1,
                                            const float* scale, const float* shift, const float* mask0,
                                            const float* mask1, int n0, int h0, int w0, int n1, int h1, int w1, int cin,
                                            int cout, long in_pix_stride, long out0_pix_stride, long out1_pix_stride,
                                            long mask0_pix_stride, long mask1_pix_stride, int flags, void* workspace,
                                            size_t workspace_bytes, dana_stream_t stream) {
  const char* who = "dana_conv3x3_winograd4_nhwc_dual";
  const long ldm0 = mask0_pix_stride > 0 ? mask0_pix_stride : cout, ldm1 = mask1_pix_stride > 0 ? mask1_pix_stride : cout;
  DANA_CHECK_ARG((!mask0 || (ldm0 % 4 == 0 && ((uintptr_t)mask0 & 15) == 0)) && (!mask1 || (ldm1 % 4 == 0 && ((uintptr_t)mask1 & 15) == 0)),
                 "%s: mask rows must be 16-byte aligned", who);
  DANA_CHECK_ARG(n0 > 0 && n1 > 0 && h0 > 0 && w0 > 0 && h1 > 0 && w1 > 0 && cin > 0 && cout > 0 && cin % 4 == 0 && cout % 4 == 0,
                 "%s: bad shape", who);
  DANA_CHECK_ARG(input && u && out0 && out1, "%s: null pointer", who);
  const long lda = in_pix_stride > 0 ? in_pix_stride : cin;
  const long ldc0 = out0_pix_stride > 0 ? out0_pix_stride : cout, ldc1 = out1_pix_stride > 0 ? out1_pix_stride : cout;
  DANA_CHECK_ARG(lda % 4 == 0 && ldc0 % 4 == 0 && ldc1 % 4 == 0 && ((uintptr_t)input & 15) == 0 && ((uintptr_t)out0 & 15) == 0 &&
                     ((uintptr_t)out1 & 15) == 0, "%s: strides / pointers must be 16-byte aligned", who);
  const long in0_bytes = (long)n0 * h0 * w0 * lda * 4, in1_bytes = (long)n1 * h1 * w1 * lda * 4;
  DANA_CHECK_ARG(in0_bytes < (long)OOB && in1_bytes < (long)OOB, "%s: input span >= 2 GiB; split the batch", who);
  const WinoPlan p0 = wino_plan(n0, h0, w0, cin, cout, 4), p1 = wino_plan(n1, h1, w1, cin, cout, 4);
  const long T = p0.tiles + p1.tiles;
  const size_t v_bytes = dana_align_up((size_t)36 * T * cin * 4, 256), need = dana_conv3x3_winograd4_dual_workspace_bytes(n0, h0, w0, n1, h1, w1, cin, cout);
  if (!workspace || workspace_bytes < need) {
    dana_set_error("%s: workspace %zu < %zu", who, workspace_bytes, need);
    return DANA_ERR_WORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  float* V = (float*)workspace;                     // [36][T][cin]: group 0's tiles, then group 1's
  float* M = (float*)((char*)workspace + v_bytes);  // [36][T][cout]
  const int C4 = cin / 4, N4 = cout / 4;
  DANA_CHECK_ARG(T * (long)(C4 > N4 ? C4 : N4) < (1L << 31), "%s: too many tiles x channels", who);
  const float* in1 = input + (long)n0 * h0 * w0 * lda;
  wino4_input_kernel<<<dana_ceil_div(p0.tiles * C4, 256), 256, 0, s>>>(input, V, h0, w0, C4, p0.th, p0.tw, p0.tiles, (int)lda,
                                                                      (unsigned)in0_bytes, T, 0);
  wino4_input_kernel<<<dana_ceil_div(p1.tiles * C4, 256), 256, 0, s>>>(in1, V, h1, w1, C4, p1.th, p1.tw, p1.tiles, (int)lda,
                                                                      (unsigned)in1_bytes, T, p0.tiles);
  DANA_CHECK_LAUNCH("dana_conv3x3_winograd4_nhwc_dual(input transforms)");
  const bool w3 = (flags & DANA_W_SPLIT3) != 0;
  const long kp = (cin + 15) / 16 * 16;
  int rc = dana_gemm_nt(V, u, M, nullptr, nullptr, nullptr, (int)T, cout, cin, cin, w3 ? kp : cin, cout, 0, 36, T * cin,
                        w3 ? 3 * cout * kp : (long)cout * cin, T * cout, 1.f, w3 ? DANA_W_SPLIT3 : 0, stream);
  if (rc) return rc;
  const int relu = (flags & DANA_EPI_RELU) ? 1 : 0;
  wino4_output_kernel<<<dana_ceil_div(p0.tiles * N4, 256), 256, 0, s>>>(M, out0, scale, shift, mask0, ldm0, h0, w0, N4, p0.th,
                                                                       p0.tw, p0.tiles, ldc0, relu, T, 0);
  wino4_output_kernel<<<dana_ceil_div(p1.tiles * N4, 256), 256, 0, s>>>(M, out1, scale, shift, mask1, ldm1, h1, w1, N4, p1.th,
                                                                       p1.tw, p1.tiles, ldc1, relu, T, p0.tiles);
  DANA_CHECK_LAUNCH("dana_conv3x3_winograd4_nhwc_dual(output transforms)");
  return DANA_OK;
}

}  // extern "C"
