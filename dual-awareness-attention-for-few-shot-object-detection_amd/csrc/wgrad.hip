// Weight gradient of the NHWC convolutions / Linear layers on the fp32 matrix cores (groundwork for the
// training step, SURVEY.md 8d variant S): the contraction runs over PIXELS,
//
//   dW[n][k] = sum_m dY[m][n] * X(m, k)          n = out channel, k = (kh, kw, cin), m = output pixel
//
// i.e. a "TN" GEMM whose reduction index is the row index of both operands. Output tile 64(n) x 64(k) per
// workgroup (4 waves 2x2, one 32x32x2 f32 MFMA tile each), reduction in steps of 32 pixels: the dY slab
// [32][64 n] and the implicit-im2col X slab [32][64 k] are staged with float4 loads along their contiguous
// (channel) axis, exactly as they sit in HBM, and the MFMA fragments are read column-wise from LDS
// (ds_read_b32, 68-dword rows: conflict-free). The pixel range is split over gridDim.z workgroups whose
// partial tiles go to a workspace and are summed in a fixed order by a second kernel (deterministic, no
// float atomics -- unlike the reference's cuDNN backward-filter algorithms).
//
// Replaces what autograd + cuDNN do for nn.Conv2d / nn.Linear weights in the reference's train step
// (train.py:141-143: loss.backward()); BN is frozen (dana.py:362-385) so dY arrives already scaled.
#include <cstdlib>

#include "common.h"
#include "../../include/dana_hip.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned OOB = 0x80000000u;
constexpr int WLD = 68;  // LDS row stride (dwords) of a [32][64] slab

struct WgradParams {
  const float* dY;   // [M][ldy]
  const float* X;    // NHWC input, pixel stride ldx
  float* partial;    // [S][N][K]
  int M, N, K;
  int IH, IW, OH, OW, Cin, KH, KW, stride, pad;
  int ldy, ldx;
  int m_chunk;       // pixels per gridDim.z slice (multiple of 32)
  int nsplit;        // gridDim.z = planes * nsplit: z = plane * nsplit + slice
  long batch_y, batch_x;  // element strides between the planes of a batched launch (Winograd-domain weight gradient)
  unsigned y_bytes, x_bytes;
};

__device__ __forceinline__ float4 ldg_b128(__amdgpu_buffer_rsrc_t r, unsigned off) {
  u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
  return *(float4*)&v;
}

// grid = (N/64 tiles, K/64 tiles, S); block 256
__global__ void __launch_bounds__(256, 2) wgrad_f32_kernel(WgradParams p) {
  __shared__ __attribute__((aligned(16))) float Gs[2][32][WLD];
  __shared__ __attribute__((aligned(16))) float Xs[2][32][WLD];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wn = wave >> 1, wk = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  const int n0 = blockIdx.x * 64, k0 = blockIdx.y * 64;
  const int plane = blockIdx.z / p.nsplit, slice = blockIdx.z - plane * p.nsplit;
  const int m_begin = slice * p.m_chunk;
  const int m_end = min(p.M, m_begin + p.m_chunk);
  const __amdgpu_buffer_rsrc_t ysrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)(p.dY + plane * p.batch_y), 0, (int)p.y_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t xsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)(p.X + plane * p.batch_x), 0, (int)p.x_bytes, 0x00020000);

  // the 64 k-columns of this tile lie inside ONE filter tap (Cin % 64 == 0): wave-uniform tap geometry
  const int tap = k0 / p.Cin, cin0 = k0 - tap * p.Cin;
  const int kh = tap / p.KW, kw = tap - kh * p.KW;
  // staging: thread -> (row r = tid / 16 (+16), float4 column c4 = tid % 16) of a [32][64] slab
  const int c4 = tid & 15, r0 = tid >> 4;
  const bool n_ok = (n0 + c4 * 4) < p.N;  // N % 4 == 0: a float4 is all-in or all-out

  float4 gy[2], gx[2];
  auto load_slab = [&](int mb) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int m = mb + r0 + 16 * j;
      const bool ok = m < m_end;
      gy[j] = ldg_b128(ysrc, (ok && n_ok) ? (unsigned)((m * p.ldy + n0 + c4 * 4) * 4) : OOB);
      const int mm = ok ? m : 0;
      const int ohw = p.OH * p.OW;
      const int img = mm / ohw, rem = mm - img * ohw;
      const int oh = rem / p.OW, ow = rem - oh * p.OW;
      const int ih = oh * p.stride - p.pad + kh, iw = ow * p.stride - p.pad + kw;
      const bool in = ok && ih >= 0 && ih < p.IH && iw >= 0 && iw < p.IW;
      gx[j] = ldg_b128(xsrc, in ? (unsigned)((((img * p.IH + ih) * p.IW + iw) * p.ldx + cin0 + c4 * 4) * 4) : OOB);
    }
  };
  auto store_slab = [&](int buf) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      *(float4*)&Gs[buf][r0 + 16 * j][c4 * 4] = gy[j];
      *(float4*)&Xs[buf][r0 + 16 * j][c4 * 4] = gx[j];
    }
  };

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  const int steps = (m_end - m_begin + 31) / 32;
  if (steps > 0) {
    load_slab(m_begin);
    store_slab(0);
    __syncthreads();
    for (int s = 0; s < steps; ++s) {
      const int buf = s & 1;
      if (s + 1 < steps) load_slab(m_begin + (s + 1) * 32);
      const float* g = &Gs[buf][lh][wn * 32 + li];
      const float* x = &Xs[buf][lh][wk * 32 + li];
#pragma unroll
      for (int q = 0; q < 16; ++q)  // MFMA step q contracts pixels {2q, 2q+1} of the slab
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(g[2 * q * WLD], x[2 * q * WLD], acc, 0, 0, 0);
      if (s + 1 < steps) store_slab(buf ^ 1);
      __syncthreads();
    }
  }
  // C/D map: col = lane&31 (k), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (n)
  float* out = p.partial + (long)blockIdx.z * p.N * p.K;
  const int k = k0 + wk * 32 + li;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int n = n0 + wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
    if (n < p.N && k < p.K) out[(long)n * p.K + k] = acc[r];
  }
}

// 128(n) x 128(k) output tile: 4 waves 2x2, each wave a 64x64 tile = 2x2 MFMA tiles, so every LDS fragment feeds two
// MFMAs (1 ds_read per MFMA instead of 2: the 64x64 kernel above sits exactly at the LDS-bandwidth limit) and the
// operands are re-read from L2 half as often. The 128 k-columns may straddle two filter taps (Cin % 64 == 0 only):
// the tap geometry is per 64-column half, i.e. per staging thread.
constexpr int WLD2 = 132;
__global__ void __launch_bounds__(256, 2) wgrad_f32_128_kernel(WgradParams p) {
  __shared__ __attribute__((aligned(16))) float Gs[2][32][WLD2];
  __shared__ __attribute__((aligned(16))) float Xs[2][32][WLD2];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wn = wave >> 1, wk = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  const int n0 = blockIdx.x * 128, k0 = blockIdx.y * 128;
  const int plane = blockIdx.z / p.nsplit, slice = blockIdx.z - plane * p.nsplit;
  const int m_begin = slice * p.m_chunk;
  const int m_end = min(p.M, m_begin + p.m_chunk);
  const __amdgpu_buffer_rsrc_t ysrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)(p.dY + plane * p.batch_y), 0, (int)p.y_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t xsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)(p.X + plane * p.batch_x), 0, (int)p.x_bytes, 0x00020000);
  // staging: thread -> rows r0 + 8 j (j = 0..3), float4 column c4 = tid % 32 of a [32][128] slab
  const int c4 = tid & 31, r0 = tid >> 5;
  const bool n_ok = (n0 + c4 * 4) < p.N;
  const int kcol = k0 + c4 * 4;
  const bool k_ok = kcol < p.K;
  const int tap = kcol / p.Cin, cin_off = kcol - tap * p.Cin;
  const int kh = tap / p.KW, kw = tap - kh * p.KW;
  const int ohw = p.OH * p.OW;

  float4 gy[4], gx[4];
  auto load_slab = [&](int mb) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = mb + r0 + 8 * j;
      const bool ok = m < m_end;
      gy[j] = ldg_b128(ysrc, (ok && n_ok) ? (unsigned)((m * p.ldy + n0 + c4 * 4) * 4) : OOB);
      const int mm = ok ? m : 0;
      const int img = mm / ohw, rem = mm - img * ohw;
      const int oh = rem / p.OW, ow = rem - oh * p.OW;
      const int ih = oh * p.stride - p.pad + kh, iw = ow * p.stride - p.pad + kw;
      const bool in = ok && k_ok && ih >= 0 && ih < p.IH && iw >= 0 && iw < p.IW;
      gx[j] = ldg_b128(xsrc, in ? (unsigned)((((img * p.IH + ih) * p.IW + iw) * p.ldx + cin_off) * 4) : OOB);
    }
  };
  auto store_slab = [&](int buf) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      *(float4*)&Gs[buf][r0 + 8 * j][c4 * 4] = gy[j];
      *(float4*)&Xs[buf][r0 + 8 * j][c4 * 4] = gx[j];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int steps = (m_end - m_begin + 31) / 32;
  if (steps > 0) {
    load_slab(m_begin);
    store_slab(0);
    __syncthreads();
    for (int s = 0; s < steps; ++s) {
      const int buf = s & 1;
      if (s + 1 < steps) load_slab(m_begin + (s + 1) * 32);
      const float* g = &Gs[buf][lh][wn * 64 + li];
      const float* x = &Xs[buf][lh][wk * 64 + li];
#pragma unroll
      for (int q = 0; q < 16; ++q) {  // MFMA step q contracts pixels {2q, 2q+1} of the slab
        const float g0 = g[2 * q * WLD2], g1 = g[2 * q * WLD2 + 32];
        const float x0 = x[2 * q * WLD2], x1 = x[2 * q * WLD2 + 32];
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(g0, x0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(g0, x1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(g1, x0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(g1, x1, acc[1][1], 0, 0, 0);
      }
      if (s + 1 < steps) store_slab(buf ^ 1);
      __syncthreads();
    }
  }
  float* out = p.partial + (long)blockIdx.z * p.N * p.K;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int k = k0 + wk * 64 + j * 32 + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + wn * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (n < p.N && k < p.K) out[(long)n * p.K + k] = acc[i][j][r];
      }
    }
}

// ---- the same 128 x 128 tile on the bf16 matrix cores (exact three-way split of the fp32 operands, six products, fp32
// accumulation: see igemm.hip) -------------------------------------------------------------------------------------
// The contraction index (pixels) is the ROW index of both operands in HBM, while the 32x32x16 MFMA wants 8 consecutive
// contraction values per lane. The transpose happens in registers on the way into LDS: a staging thread loads a
// 4 (pixels) x 4 (channels) block -- four float4 along the contiguous channel axis, like the f32 kernel -- splits the 16
// values and writes, per channel and plane, the four pixel-consecutive bf16 as one ds_write_b64 into that channel's LDS
// row (rows of 16 bf16 padded to 48 B, the igemm_split layout, so the fragments are plain ds_read_b128). Waves 0-1
// stage dY, waves 2-3 the implicit-im2col X slab. Channel c of the tile lives in LDS row (c & 3) * 32 + (c >> 2):
// consecutive lanes write consecutive rows (no 8-way bank conflict), and the epilogue undoes the permutation.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int WSLD = 12;  // dwords per LDS row of one plane
constexpr int WSK = 16;   // pixels per step

__global__ void __launch_bounds__(256, 2) wgrad_split_128_kernel(WgradParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned wsm[];  // [2 operands][2 buffers][3 planes][128 rows][WSLD]
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wn = wave >> 1, wk = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  const int n0 = blockIdx.x * 128, k0 = blockIdx.y * 128;
  const int plane = blockIdx.z / p.nsplit, slice = blockIdx.z - plane * p.nsplit;
  const int m_begin = slice * p.m_chunk;
  const int m_end = min(p.M, m_begin + p.m_chunk);
  const int op = tid >> 7;                  // 0: dY (waves 0, 1), 1: X (waves 2, 3) -- wave-uniform
  const int cq = tid & 31, mg = (tid >> 5) & 3;  // column quad, pixel group of 4
  const __amdgpu_buffer_rsrc_t src = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(op ? p.X + plane * p.batch_x : p.dY + plane * p.batch_y), 0, (int)(op ? p.x_bytes : p.y_bytes), 0x00020000);
  unsigned* const stage = wsm + op * (2 * 3 * 128 * WSLD) + cq * WSLD + mg * 2;  // + buf*3*128*WSLD + (plane*128 + e*32)*WSLD

  // per-thread geometry of its four rows (pixels m_begin + 4 mg + i, advancing by 16 per step)
  const int col = (op ? k0 : n0) + cq * 4;
  const bool col_ok = col < (op ? p.K : p.N);
  const int tap = op ? col / p.Cin : 0, cin_off = op ? col - tap * p.Cin : col;
  const int kh = tap / p.KW, kw = tap - kh * p.KW;
  const bool plain = !op || (p.KH * p.KW == 1 && p.stride == 1 && p.pad == 0);  // row m of the operand is pixel m
  const int ld4 = (op ? p.ldx : p.ldy) * 4;
  int m_row[4], img[4], oh[4], ow[4];
  const int ohw = p.OH * p.OW;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m_begin + mg * 4 + i;
    m_row[i] = m;
    img[i] = oh[i] = ow[i] = 0;
    if (!plain) {  // (wave-uniform: the dY half and every 1x1 / stride-1 X half address row m directly, no divisions)
      const int mm = m < p.M ? m : 0;
      img[i] = mm / ohw;
      const int rem = mm - img[i] * ohw;
      oh[i] = rem / p.OW;
      ow[i] = rem - oh[i] * p.OW;
    }
  }
  int d_ow = 0, d_oh = 0, d_img = 0;
  if (!plain) {
    d_ow = WSK % p.OW;
    d_oh = (WSK / p.OW) % p.OH;
    d_img = WSK / ohw;
  }

  float4 r[4];
  auto load_slab = [&]() {  // the thread's 4 x 4 block of the next 16 pixels, then advance
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      unsigned off;
      bool ok = col_ok && m_row[i] < m_end;
      if (plain) {
        off = (unsigned)(m_row[i] * ld4 + cin_off * 4);
      } else {
        const int ih = oh[i] * p.stride - p.pad + kh, iw = ow[i] * p.stride - p.pad + kw;
        ok = ok && ih >= 0 && ih < p.IH && iw >= 0 && iw < p.IW;
        off = (unsigned)(((img[i] * p.IH + ih) * p.IW + iw) * ld4 + cin_off * 4);
        ow[i] += d_ow;
        const int c1 = ow[i] >= p.OW ? 1 : 0;
        ow[i] -= c1 ? p.OW : 0;
        oh[i] += d_oh + c1;
        const int c2 = oh[i] >= p.OH ? 1 : 0;
        oh[i] -= c2 ? p.OH : 0;
        img[i] += d_img + c2;
      }
      r[i] = ldg_b128(src, ok ? off : OOB);
      m_row[i] += WSK;
    }
  };
  auto store_slab = [&](int buf) {
    unsigned* w = stage + buf * (3 * 128 * WSLD);
#pragma unroll
    for (int e = 0; e < 4; ++e) {  // channel e of the quad: its four pixels -> 4 bf16 per plane
      unsigned hb[4], mb[4], lb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float x = e == 0 ? r[i].x : e == 1 ? r[i].y : e == 2 ? r[i].z : r[i].w;
        hb[i] = __float_as_uint(x) & 0xffff0000u;
        const float r1 = x - __uint_as_float(hb[i]);  // exact
        mb[i] = __float_as_uint(r1) & 0xffff0000u;
        lb[i] = __float_as_uint(r1 - __uint_as_float(mb[i]));  // exact; <= 8 significant bits
      }
      uint2 h, m, l;
      h.x = __builtin_amdgcn_perm(hb[1], hb[0], 0x07060302u);
      h.y = __builtin_amdgcn_perm(hb[3], hb[2], 0x07060302u);
      m.x = __builtin_amdgcn_perm(mb[1], mb[0], 0x07060302u);
      m.y = __builtin_amdgcn_perm(mb[3], mb[2], 0x07060302u);
      l.x = __builtin_amdgcn_perm(lb[1], lb[0], 0x07060302u);
      l.y = __builtin_amdgcn_perm(lb[3], lb[2], 0x07060302u);
      *(uint2*)(w + (0 * 128 + e * 32) * WSLD) = h;
      *(uint2*)(w + (1 * 128 + e * 32) * WSLD) = m;
      *(uint2*)(w + (2 * 128 + e * 32) * WSLD) = l;
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

  const int steps = (m_end - m_begin + WSK - 1) / WSK;
  if (steps > 0) {
    load_slab();
    store_slab(0);
    __syncthreads();
    load_slab();
    for (int s = 0; s < steps; ++s) {
      const int buf = s & 1;
      const unsigned* g = wsm + buf * (3 * 128 * WSLD) + (wn * 64 + li) * WSLD + lh * 4;
      const unsigned* x = wsm + (2 + buf) * (3 * 128 * WSLD) + (wk * 64 + li) * WSLD + lh * 4;
      u32x4 gf[3][2], xf[3][2];
#pragma unroll
      for (int pc = 0; pc < 3; ++pc)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          gf[pc][i] = *(const u32x4*)(g + (pc * 128 + i * 32) * WSLD);
          xf[pc][i] = *(const u32x4*)(x + (pc * 128 + i * 32) * WSLD);
        }
      constexpr int PA[6] = {0, 0, 1, 1, 0, 2};
      constexpr int PB[6] = {0, 1, 0, 1, 2, 0};
#pragma unroll
      for (int q = 0; q < 6; ++q)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, gf[PA[q]][i]),
                                                               __builtin_bit_cast(bf16x8, xf[PB[q]][j]), acc[i][j], 0, 0, 0);
      store_slab(buf ^ 1);  // pixels of step s + 1 (zeros past the end)
      load_slab();          // step s + 2
      __syncthreads();
    }
  }
  // C/D map: col = lane&31, row = (q&3) + 8*(q>>2) + 4*(lane>>5) within a 32x32 tile; LDS row rho <-> channel
  // 4*(rho & 31) + (rho >> 5) of the 128-wide tile
  float* out = p.partial + (long)blockIdx.z * p.N * p.K;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int rk = wk * 64 + j * 32 + li;
      const int k = k0 + 4 * (rk & 31) + (rk >> 5);
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int rn = wn * 64 + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * lh;
        const int n = n0 + 4 * (rn & 31) + (rn >> 5);
        if (n < p.N && k < p.K) out[(long)n * p.K + k] = acc[i][j][q];
      }
    }
}

// ---- the same tile and the same LDS image, software-pipelined like igemm_split_kernel (igemm.hip): three stages in
// flight (global -> registers one step ahead, registers -> split -> LDS planes, LDS -> fragments one step ahead), every
// non-MFMA instruction of a step cut into micro-items of two independent instructions and dealt out over the 24 MFMAs,
// the step's one barrier behind the fourth MFMA. For the launches whose operand rows ARE pixel rows (every 1x1 /
// stride-1 conv, every Linear, the Winograd-domain batched planes): no per-load geometry, one compare + select + add per
// load. The loop above (compiler-scheduled: fragments read at the top of the step, the split behind the MFMAs) stays
// for the strided / multi-tap launches.
template <int V>
struct WIC {
  static constexpr int value = V;
};
template <int I, int N, class F>
__device__ __forceinline__ void wstatic_for(F&& f) {
  if constexpr (I < N) {
    f(WIC<I>{});
    wstatic_for<I + 1, N>(f);
  }
}
struct WStepSched {  // (make_step_sched of igemm.hip for NM = 24, four loads, twelve reads, four split units)
  static constexpr int NM = 24, NL = 4, NR = 12, NU = 4, NCV = 13 * NU, NLH = 2 * NL, NIT = NCV + NR + NLH, SB = 4;
  int kind[NIT], idx[NIT], gap[NIT];
};
constexpr WStepSched make_wstep_sched() {
  WStepSched s{};
  int n = 0;
  auto put = [&](int kind, int idx, long t, bool lds) {
    int g = (int)(t * WStepSched::NM / 10000);
    g = g < WStepSched::NM ? g : WStepSched::NM - 1;
    if (lds && g < WStepSched::SB) g = WStepSched::SB;
    s.kind[n] = kind;
    s.idx[n] = idx;
    s.gap[n++] = g;
  };
  for (int l = 0; l < WStepSched::NLH; ++l) put(0, l, (2 * l + 1) * 3000 / (2 * WStepSched::NLH), false);
  for (int r = 0; r < WStepSched::NR; ++r) put(1, r, 1700 + (2 * r + 1) * 8000 / (2 * WStepSched::NR), true);
  for (int i = 0; i < WStepSched::NCV; ++i) put(2, i, (2 * i + 1) * 9950 / (2 * WStepSched::NCV), i % 13 >= 11);
  return s;
}
inline constexpr WStepSched kWStepSched = make_wstep_sched();

__global__ void __launch_bounds__(256, 2) wgrad_split_128p_kernel(WgradParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned wsm[];  // [2 operands][2 buffers][3 planes][128 rows][WSLD]
  using SCH = WStepSched;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wn = wave >> 1, wk = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  const int n0 = blockIdx.x * 128, k0 = blockIdx.y * 128;
  const int plane = blockIdx.z / p.nsplit, slice = blockIdx.z - plane * p.nsplit;
  const int m_begin = slice * p.m_chunk;
  const int m_end = min(p.M, m_begin + p.m_chunk);
  const int op = tid >> 7;                       // 0: dY (waves 0, 1), 1: X (waves 2, 3) -- wave-uniform
  const int cq = tid & 31, mg = (tid >> 5) & 3;  // column quad, pixel group of 4
  const __amdgpu_buffer_rsrc_t src = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(op ? p.X + plane * p.batch_x : p.dY + plane * p.batch_y), 0, (int)(op ? p.x_bytes : p.y_bytes), 0x00020000);
  unsigned* const stage = wsm + op * (2 * 3 * 128 * WSLD) + cq * WSLD + mg * 2;
  const int col = (op ? k0 : n0) + cq * 4;
  const bool col_ok = col < (op ? p.K : p.N);
  const int ld4 = (op ? p.ldx : p.ldy) * 4;
  constexpr int DEAD = (int)0x80000000;
  // row m_begin + 4 mg + i of the operand (advancing by 16 per step); a row is live while its pixel index < lim
  unsigned cur[4];
  int mrow[4];
  const int lim = col_ok ? m_end : DEAD;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    mrow[i] = m_begin + mg * 4 + i;
    cur[i] = (unsigned)(mrow[i] * ld4 + col * 4);
  }
  const unsigned step_bytes = (unsigned)(WSK * ld4);
  auto load_one = [&](int i, float4& dst) {
    unsigned off = mrow[i] < lim ? cur[i] : OOB;
    asm volatile("" : "+v"(off));
    dst = ldg_b128(src, off);
    cur[i] += step_bytes;
    mrow[i] += WSK;
  };
  float4 r0[4], r1[4];
  auto load_slab = [&](float4(&r)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) load_one(i, r[i]);
  };
  auto store_slab = [&](int buf, const float4(&r)[4]) {
    unsigned* w = stage + buf * (3 * 128 * WSLD);
#pragma unroll
    for (int e = 0; e < 4; ++e) {  // channel e of the quad: its four pixels -> 4 bf16 per plane
      unsigned hb[4], mb[4], lb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float x = e == 0 ? r[i].x : e == 1 ? r[i].y : e == 2 ? r[i].z : r[i].w;
        hb[i] = __float_as_uint(x) & 0xffff0000u;
        const float t1 = x - __uint_as_float(hb[i]);
        mb[i] = __float_as_uint(t1) & 0xffff0000u;
        lb[i] = __float_as_uint(t1 - __uint_as_float(mb[i]));
      }
      uint2 h, m, l;
      h.x = __builtin_amdgcn_perm(hb[1], hb[0], 0x07060302u);
      h.y = __builtin_amdgcn_perm(hb[3], hb[2], 0x07060302u);
      m.x = __builtin_amdgcn_perm(mb[1], mb[0], 0x07060302u);
      m.y = __builtin_amdgcn_perm(mb[3], mb[2], 0x07060302u);
      l.x = __builtin_amdgcn_perm(lb[1], lb[0], 0x07060302u);
      l.y = __builtin_amdgcn_perm(lb[3], lb[2], 0x07060302u);
      *(uint2*)(w + (0 * 128 + e * 32) * WSLD) = h;
      *(uint2*)(w + (1 * 128 + e * 32) * WSLD) = m;
      *(uint2*)(w + (2 * 128 + e * 32) * WSLD) = l;
    }
  };
  // fragments: [plane][0..1] = dY rows wn*64 + {0, 32} + li, [plane][2..3] = X rows wk*64 + {0, 32} + li
  u32x4 f0[3][4], f1[3][4];
  auto frag_ptr = [&](int buf, int x) {
    return x < 2 ? wsm + buf * (3 * 128 * WSLD) + (wn * 64 + x * 32 + li) * WSLD + lh * 4
                 : wsm + (2 + buf) * (3 * 128 * WSLD) + (wk * 64 + (x - 2) * 32 + li) * WSLD + lh * 4;
  };
  auto read_frags = [&](int buf, u32x4(&f)[3][4]) {
#pragma unroll
    for (int pc = 0; pc < 3; ++pc)
#pragma unroll
      for (int x = 0; x < 4; ++x) f[pc][x] = *(const u32x4*)(frag_ptr(buf, x) + pc * 128 * WSLD);
  };

  f32x16 acc[2][2];
  const int steps = (m_end - m_begin + WSK - 1) / WSK;
  load_slab(r0);  // tile 0
  load_slab(r1);  // tile 1
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
      asm volatile("" : "+a"(acc[i][j]));
    }
  __builtin_amdgcn_sched_barrier(0);
  store_slab(0, r0);
  __syncthreads();
  load_slab(r0);  // tile 2
  read_frags(0, f0);
  store_slab(1, r1);
  __syncthreads();

  // step t: MFMAs on tile t (f), reads tile t+1 into nf, splits tile t+2 (cv) into LDS[t & 1], requests tile t+3 (ld)
  auto k_step = [&](int par, const u32x4(&f)[3][4], u32x4(&nf)[3][4], const float4(&cv)[4], float4(&ld)[4]) {
    const int bw_ = par, br_ = par ^ 1;
    unsigned* w = stage + bw_ * (3 * 128 * WSLD);
    constexpr int PA[6] = {0, 0, 1, 1, 0, 2};
    constexpr int PB[6] = {0, 1, 0, 1, 2, 0};
    unsigned hb[4][4], mb[4][4], lb[4][4], ld_off[4];
    float t1[4][4];
    uint2 hp[4], mp[4], lp[4];
    __builtin_amdgcn_sched_barrier(0);
    wstatic_for<0, SCH::NM>([&](auto qc) {
      constexpr int q = decltype(qc)::value;
      constexpr int pq = q / 4, ti = (q / 2) % 2, tj = q % 2;
      acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f[PA[pq]][ti]),
                                                            __builtin_bit_cast(bf16x8, f[PB[pq]][2 + tj]), acc[ti][tj], 0, 0, 0);
      asm volatile("" : "+a"(acc[ti][tj]));
      if constexpr (q == SCH::SB) __syncthreads();
      wstatic_for<0, SCH::NIT>([&](auto ic) {
        constexpr int it = decltype(ic)::value;
        constexpr int kind = kWStepSched.kind[it], ix = kWStepSched.idx[it];
        if constexpr (kWStepSched.gap[it] != q) {
        } else if constexpr (kind == 0) {
          constexpr int i = ix / 2, half = ix % 2;
          if constexpr (half == 0) {
            unsigned off = mrow[i] < lim ? cur[i] : OOB;
            asm volatile("" : "+v"(off));
            ld_off[i] = off;
          } else {
            ld[i] = ldg_b128(src, ld_off[i]);
            cur[i] += step_bytes;
            mrow[i] += WSK;
          }
        } else if constexpr (kind == 1) {
          constexpr int pc = ix / 4, x = ix % 4;
          nf[pc][x] = *(const u32x4*)(frag_ptr(br_, x) + pc * 128 * WSLD);
        } else {
          constexpr int e = ix / 13, r = ix % 13;  // unit e = channel e of the quad, its elements = the four pixels
          constexpr int i0 = (r & 1) * 2, i1 = i0 + 1;
          auto el = [&](int i) { return e == 0 ? cv[i].x : e == 1 ? cv[i].y : e == 2 ? cv[i].z : cv[i].w; };
          if constexpr (r < 2) {
            hb[e][i0] = __float_as_uint(el(i0)) & 0xffff0000u;
            hb[e][i1] = __float_as_uint(el(i1)) & 0xffff0000u;
            asm volatile("" : "+v"(hb[e][i0]), "+v"(hb[e][i1]));
          } else if constexpr (r < 4) {
            t1[e][i0] = el(i0) - __uint_as_float(hb[e][i0]);
            t1[e][i1] = el(i1) - __uint_as_float(hb[e][i1]);
            asm volatile("" : "+v"(t1[e][i0]), "+v"(t1[e][i1]));
          } else if constexpr (r < 6) {
            mb[e][i0] = __float_as_uint(t1[e][i0]) & 0xffff0000u;
            mb[e][i1] = __float_as_uint(t1[e][i1]) & 0xffff0000u;
            asm volatile("" : "+v"(mb[e][i0]), "+v"(mb[e][i1]));
          } else if constexpr (r < 8) {
            lb[e][i0] = __float_as_uint(t1[e][i0] - __uint_as_float(mb[e][i0]));
            lb[e][i1] = __float_as_uint(t1[e][i1] - __uint_as_float(mb[e][i1]));
            asm volatile("" : "+v"(lb[e][i0]), "+v"(lb[e][i1]));
          } else if constexpr (r == 8) {
            hp[e].x = __builtin_amdgcn_perm(hb[e][1], hb[e][0], 0x07060302u);
            hp[e].y = __builtin_amdgcn_perm(hb[e][3], hb[e][2], 0x07060302u);
            asm volatile("" : "+v"(hp[e].x), "+v"(hp[e].y));
          } else if constexpr (r == 9) {
            mp[e].x = __builtin_amdgcn_perm(mb[e][1], mb[e][0], 0x07060302u);
            mp[e].y = __builtin_amdgcn_perm(mb[e][3], mb[e][2], 0x07060302u);
            asm volatile("" : "+v"(mp[e].x), "+v"(mp[e].y));
          } else if constexpr (r == 10) {
            lp[e].x = __builtin_amdgcn_perm(lb[e][1], lb[e][0], 0x07060302u);
            lp[e].y = __builtin_amdgcn_perm(lb[e][3], lb[e][2], 0x07060302u);
            asm volatile("" : "+v"(lp[e].x), "+v"(lp[e].y));
          } else if constexpr (r == 11) {
            *(uint2*)(w + (0 * 128 + e * 32) * WSLD) = hp[e];
            *(uint2*)(w + (1 * 128 + e * 32) * WSLD) = mp[e];
          } else {
            *(uint2*)(w + (2 * 128 + e * 32) * WSLD) = lp[e];
          }
        }
      });
      __builtin_amdgcn_sched_barrier(0);
    });
  };
  for (int t = 0; t < steps; t += 2) {  // (whole pairs: an odd count runs one all-zero step)
    k_step(0, f0, f1, r0, r1);
    k_step(1, f1, f0, r1, r0);
  }
  // C/D map: col = lane&31, row = (q&3) + 8*(q>>2) + 4*(lane>>5) within a 32x32 tile; LDS row rho <-> channel
  // 4*(rho & 31) + (rho >> 5) of the 128-wide tile
  float* out = p.partial + (long)blockIdx.z * p.N * p.K;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int rk = wk * 64 + j * 32 + li;
      const int k = k0 + 4 * (rk & 31) + (rk >> 5);
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int rn = wn * 64 + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * lh;
        const int n = n0 + 4 * (rn & 31) + (rn >> 5);
        if (n < p.N && k < p.K) out[(long)n * p.K + k] = acc[i][j][q];
      }
    }
}

// dW[i] = (accumulate ? dW[i] : 0) + row_scale[row(i)] * sum_s partial[s][i], fixed order. row_scale (optional) is the
// frozen-BN scale of the output channel: with it the launch writes straight into the parameter's gradient.
// A block = 64 float4 outputs x 4 slice-lanes (lane q sums slices q, q+4, ...; the four are combined in a fixed order
// through LDS): 4x the blocks and S/4 dependent loads per thread -- the 1-lane version was latency-bound (~19 us).
__global__ void __launch_bounds__(256)
wgrad_reduce_kernel(const float4* __restrict__ partial, float4* __restrict__ dW, const float* __restrict__ row_scale,
                    long n4, int K4, int S, int accumulate, long out_plane4 = -1, long n4_valid = -1) {
  __shared__ float4 part[4][64];
  const int ol = threadIdx.x & 63, q = threadIdx.x >> 6;
  const long i = (long)blockIdx.x * 64 + ol;
  partial += (long)blockIdx.y * S * n4;  // plane of a batched launch
  dW += (long)blockIdx.y * (out_plane4 >= 0 ? out_plane4 : n4);
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  const long n4s = n4;  // (slice stride of the partial sums)
  if (n4_valid >= 0) n4 = n4_valid;  // only the first rows of a plane are results (zero-padded operand rows behind them)
  if (i < n4) {
#pragma unroll 4
    for (int s = q; s < S; s += 4) {
      const float4 v = partial[(long)s * n4s + i];
      a.x += v.x;
      a.y += v.y;
      a.z += v.z;
      a.w += v.w;
    }
  }
  part[q][ol] = a;
  __syncthreads();
  if (q != 0 || i >= n4) return;
  const float4 b = part[1][ol], c = part[2][ol], d = part[3][ol];
  a.x = (a.x + b.x) + (c.x + d.x);
  a.y = (a.y + b.y) + (c.y + d.y);
  a.z = (a.z + b.z) + (c.z + d.z);
  a.w = (a.w + b.w) + (c.w + d.w);
  if (row_scale) {
    const float sc = row_scale[i / K4];
    a.x *= sc;
    a.y *= sc;
    a.z *= sc;
    a.w *= sc;
  }
  if (accumulate) {
    const float4 o = dW[i];
    a.x += o.x;
    a.y += o.y;
    a.z += o.z;
    a.w += o.w;
  }
  dW[i] = a;
}

// packed weight [cout][kh][kw][cin] (* scale[cout]) -> data-gradient weight [cin][kh'][kw'][cout], spatially flipped.
// One 32x32 (cout x cin) tile of one tap per block through LDS: reads coalesced along cin, writes along cout.
__global__ void __launch_bounds__(256)
dgrad_weight_kernel(const float* __restrict__ w, const float* __restrict__ scale, float* __restrict__ out, int cout,
                    int cin, int KH, int KW) {
  __shared__ float tile[32][33];
  const int taps = KH * KW;
  const int tap = blockIdx.z, ftap = taps - 1 - tap;  // (KH-1-kh, KW-1-kw) of the source
  const int ci0 = blockIdx.x * 32, co0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
#pragma unroll
  for (int r = ty; r < 32; r += 8) {
    const int co = co0 + r, ci = ci0 + tx;
    float v = 0.f;
    if (co < cout && ci < cin) v = w[((long)co * taps + ftap) * cin + ci] * (scale ? scale[co] : 1.f);
    tile[r][tx] = v;
  }
  __syncthreads();
#pragma unroll
  for (int r = ty; r < 32; r += 8) {
    const int ci = ci0 + r, co = co0 + tx;
    if (ci < cin && co < cout) out[((long)ci * taps + tap) * cout + co] = tile[tx][r];
  }
}

// strided 1x1 data gradient: g_in[b][2oh*s][2ow*s][:] = compact[b][oh][ow][:], zeros elsewhere (memset by caller)
__global__ void __launch_bounds__(256)
upsample_scatter_kernel(const float4* __restrict__ compact, float4* __restrict__ out, const float4* __restrict__ mask,
                        int OH, int OW, int IH, int IW, int C4, int stride, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % C4);
  const int ow = (int)((i / C4) % OW);
  const int oh = (int)((i / C4 / OW) % OH);
  const long b = i / C4 / OW / OH;
  const long o = ((b * IH + (long)oh * stride) * IW + (long)ow * stride) * C4 + c;
  float4 v = compact[i];
  if (mask) {  // ReLU adjoint of the layer that produced the conv's input
    const float4 k = mask[o];
    v.x = k.x > 0.f ? v.x : 0.f;
    v.y = k.y > 0.f ? v.y : 0.f;
    v.z = k.z > 0.f ? v.z : 0.f;
    v.w = k.w > 0.f ? v.w : 0.f;
  }
  out[o] = v;
}

// the input pixels a strided 1x1 conv reads, as plain rows: compact[b][oh][ow][:] = x[b][oh*s][ow*s][:]
__global__ void __launch_bounds__(256)
downsample_gather_kernel(const float* __restrict__ x, float4* __restrict__ compact, int OH, int OW, int IH, int IW, int C4,
                         int stride, long ldx, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % C4);
  const int ow = (int)((i / C4) % OW);
  const int oh = (int)((i / C4 / OW) % OH);
  const long b = i / C4 / OW / OH;
  compact[i] = *(const float4*)(x + ((b * IH + (long)oh * stride) * IW + (long)ow * stride) * ldx + c * 4);
}

// tile edge (64 or 128) and pixel split of one weight-gradient launch
struct WgradShape {
  int tile, tn, tk, S;
};
WgradShape wgrad_shape(int N, int K, int M, int planes = 1) {
  // Pick (tile, S) by a small cost model: `rounds` of workgroups over the chip's slots (2 per CU for the 128x128
  // tile, 4 for 64x64), each doing chunk/32 slabs, plus the HBM round trip of the S partial tiles. The 128x128
  // tile sustains ~0.75 of the per-CU MFMA peak (one LDS read per MFMA), the 64x64 tile ~0.5 (two).
  WgradShape best = {64, 1, 1, 1};
  double best_t = 1e30;
  for (int tile = 64; tile <= 128; tile += 64) {
    const int tn = (N + tile - 1) / tile, tk = (K + tile - 1) / tile;
    const long tiles = (long)tn * tk * planes;
    const int slots = tile == 128 ? 512 : 1024;
    // the 128 x 128 tile runs on the bf16x6 split kernel (419.4 TFLOP/s ceiling, ~0.5 of it sustained) unless
    // dana_set_mfma_mode(0); the 64 x 64 tile is always the f32-MFMA kernel (157.3)
    const bool split = tile == 128 && dana_get_mfma_mode() != 0;
    const double cu_flops = split ? 419.4e12 / 256.0 * 0.5 : 157.3e12 / 256.0 * (tile == 128 ? 0.75 : 0.5);
    const double wg_flops = cu_flops / (tile == 128 ? 2 : 4);
    const int maxS = (M + 255) / 256 < 64 ? (M + 255) / 256 : 64;
    for (int S = 1; S <= (maxS < 1 ? 1 : maxS); ++S) {
      const int chunk = ((M + S - 1) / S + 31) / 32 * 32;
      const long rounds = (tiles * S + slots - 1) / slots;
      const double t_mma = (double)rounds * (chunk / 32) * (2.0 * tile * tile * 32) / wg_flops + rounds * 2e-6;
      const double t_part = 2.0 * S * (double)N * K * planes * 4.0 / 4e12;
      const double t = t_mma + t_part;
      if (t < best_t) {
        best_t = t;
        best = {tile, tn, tk, S};
      }
    }
  }
  return best;
}

// 128 x 128 tile: bf16x6 split kernel unless dana_set_mfma_mode(0)
void launch_wgrad_128(const WgradParams& p, dim3 grid, hipStream_t s) {
  if (dana_get_mfma_mode() != 0) {
    constexpr int lds = 2 * 2 * 3 * 128 * WSLD * (int)sizeof(unsigned);
    static DeviceOnce attr;
    attr.once([&] { return hipFuncSetAttribute((const void*)wgrad_split_128_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds); });
    // rows of both operands are pixel rows (1x1 / stride 1 / no padding: every Linear, most convs, the Winograd-domain
    // planes): the software-pipelined kernel; strided / multi-tap launches: the general one
    if (p.KH * p.KW == 1 && p.stride == 1 && p.pad == 0) {
      static DeviceOnce attr_p;
      attr_p.once([&] { return hipFuncSetAttribute((const void*)wgrad_split_128p_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds); });
      wgrad_split_128p_kernel<<<grid, 256, lds, s>>>(p);
    } else {
      wgrad_split_128_kernel<<<grid, 256, lds, s>>>(p);
    }
  } else {
    wgrad_f32_128_kernel<<<grid, 256, 0, s>>>(p);
  }
}

}  // namespace

// ---- internal (common.h): batched "TN" GEMM out[z][N][K] = dY[z]^T . X[z] over M rows, for the Winograd-domain weight
// gradient (winograd.hip). No accumulate / scale; partial slices through the workspace as above.
size_t dana_wgrad_tn_batched_workspace(int planes, int M, int N, int K) {
  const WgradShape ws = wgrad_shape(N, K, M, planes);
  return ws.S > 1 ? (size_t)planes * ws.S * N * K * sizeof(float) : 0;
}

int dana_wgrad_tn_batched(const float* dY, const float* X, float* out, int planes, int M, int N, int K, long batch_y,
                          long batch_x, void* workspace, size_t workspace_bytes, void* stream) {
  DANA_CHECK_ARG(planes > 0 && M > 0 && N > 0 && K > 0 && K % 64 == 0 && N % 4 == 0, "dana_wgrad_tn_batched: bad shape");
  WgradParams p;
  memset(&p, 0, sizeof(p));
  p.dY = dY;
  p.X = X;
  p.IH = 1;
  p.IW = M;
  p.OH = 1;
  p.OW = M;
  p.M = M;
  p.N = N;
  p.K = K;
  p.Cin = K;
  p.KH = p.KW = 1;
  p.stride = 1;
  p.pad = 0;
  p.ldx = K;
  p.ldy = N;
  p.batch_y = batch_y;
  p.batch_x = batch_x;
  const long xb = (long)M * K * 4, yb = (long)M * N * 4;
  DANA_CHECK_ARG(xb < (long)OOB && yb < (long)OOB, "dana_wgrad_tn_batched: plane >= 2 GiB");
  p.x_bytes = (unsigned)xb;
  p.y_bytes = (unsigned)yb;
  const WgradShape ws = wgrad_shape(N, K, M, planes);
  const int S = ws.S;
  p.m_chunk = ((M + S - 1) / S + 31) / 32 * 32;
  p.nsplit = S;
  const size_t need = dana_wgrad_tn_batched_workspace(planes, M, N, K);
  if (need && (!workspace || workspace_bytes < need)) {
    dana_set_error("dana_wgrad_tn_batched: workspace %zu < %zu", workspace_bytes, need);
    return DANA_ERR_WORKSPACE;
  }
  p.partial = S > 1 ? (float*)workspace : out;  // one slice: the kernel writes the result itself
  DANA_CHECK_ARG((long)planes * S <= 65535, "dana_wgrad_tn_batched: too many slices");
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(ws.tn, ws.tk, planes * S);
  if (ws.tile == 128)
    launch_wgrad_128(p, grid, s);
  else
    wgrad_f32_kernel<<<grid, 256, 0, s>>>(p);
  DANA_CHECK_LAUNCH("dana_wgrad_tn_batched");
  if (S > 1) {
    const long n4 = (long)N * K / 4;
    dim3 rgrid(dana_ceil_div(n4, 64), planes);
    wgrad_reduce_kernel<<<rgrid, 256, 0, s>>>((const float4*)workspace, (float4*)out, nullptr, n4, K / 4, S, 0);
    DANA_CHECK_LAUNCH("dana_wgrad_tn_batched(reduce)");
  }
  return DANA_OK;
}

extern "C" {

/* out[z][n][k] (+)= sum_m y[z][m][n] * x[z][m][k] for n < n_valid: a batch of "TN" GEMMs (the adjoints of torch.bmm
 * w.r.t. its right operand, dana.py:140-150 / 270-283, one plane per image). N rows are computed (y may carry zero
 * padded columns behind n_valid); k % 64 == 0, n % 4 == 0. Deterministic split-M reduction through the workspace. */
size_t dana_gemm_tn_batched_workspace_bytes(int planes, int m, int n, int k) {
  if (planes <= 0 || m <= 0 || n <= 0 || k <= 0) return 0;
  return (size_t)planes * wgrad_shape(n, k, m, planes).S * n * k * sizeof(float);
}

int dana_gemm_tn_batched(const float* y, const float* x, float* out, int planes, int m, int n, int k, long ldy, long ldx,
                         long batch_y, long batch_x, long batch_out, int n_valid, int accumulate, void* workspace,
                         size_t workspace_bytes, dana_stream_t stream) {
  DANA_CHECK_ARG(planes >= 0 && m > 0 && n > 0 && k > 0 && k % 64 == 0 && n % 4 == 0 && n_valid >= 0 && n_valid <= n,
                 "dana_gemm_tn_batched: bad shape (k %% 64 == 0, n %% 4 == 0)");
  if (planes == 0 || n_valid == 0) return DANA_OK;
  DANA_CHECK_ARG(y && x && out, "dana_gemm_tn_batched: null pointer");
  if (ldy <= 0) ldy = n;
  if (ldx <= 0) ldx = k;
  DANA_CHECK_ARG(ldy % 4 == 0 && ldx % 4 == 0 && ldy >= n && ldx >= k && batch_y % 4 == 0 && batch_x % 4 == 0 &&
                     batch_out % 4 == 0 && (((uintptr_t)y | (uintptr_t)x | (uintptr_t)out) & 15) == 0,
                 "dana_gemm_tn_batched: strides / pointers must be 16-byte aligned");
  WgradParams p;
  memset(&p, 0, sizeof(p));
  p.dY = y;
  p.X = x;
  p.IH = 1;
  p.IW = m;
  p.OH = 1;
  p.OW = m;
  p.M = m;
  p.N = n;
  p.K = k;
  p.Cin = k;
  p.KH = p.KW = 1;
  p.stride = 1;
  p.pad = 0;
  p.ldx = (int)ldx;
  p.ldy = (int)ldy;
  p.batch_y = batch_y;
  p.batch_x = batch_x;
  const long xb = (long)m * ldx * 4, yb = (long)m * ldy * 4;
  DANA_CHECK_ARG(xb < (long)OOB && yb < (long)OOB, "dana_gemm_tn_batched: plane >= 2 GiB");
  p.x_bytes = (unsigned)xb;
  p.y_bytes = (unsigned)yb;
  const WgradShape ws = wgrad_shape(n, k, m, planes);
  const int S = ws.S;
  p.m_chunk = ((m + S - 1) / S + 31) / 32 * 32;
  p.nsplit = S;
  const size_t need = (size_t)planes * S * n * k * sizeof(float);
  if (!workspace || workspace_bytes < need) {
    dana_set_error("dana_gemm_tn_batched: workspace %zu < %zu", workspace_bytes, need);
    return DANA_ERR_WORKSPACE;
  }
  p.partial = (float*)workspace;
  DANA_CHECK_ARG((long)planes * S <= 65535, "dana_gemm_tn_batched: too many slices");
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(ws.tn, ws.tk, planes * S);
  if (ws.tile == 128)
    launch_wgrad_128(p, grid, s);
  else
    wgrad_f32_kernel<<<grid, 256, 0, s>>>(p);
  DANA_CHECK_LAUNCH("dana_gemm_tn_batched");
  const long n4 = (long)n * k / 4, n4v = (long)n_valid * k / 4;
  dim3 rgrid(dana_ceil_div(n4v, 64), planes);
  wgrad_reduce_kernel<<<rgrid, 256, 0, s>>>((const float4*)workspace, (float4*)out, nullptr, n4, k / 4, S, accumulate,
                                            batch_out > 0 ? batch_out / 4 : n4v, n4v);
  DANA_CHECK_LAUNCH("dana_gemm_tn_batched(reduce)");
  return DANA_OK;
}

size_t dana_conv2d_wgrad_workspace_bytes(int batch, int in_h, int in_w, int cin, int cout, int kh, int kw, int stride,
                                         int pad) {
  if (batch <= 0 || cin <= 0 || cout <= 0) return 0;
  const int oh = (in_h + 2 * pad - kh) / stride + 1, ow = (in_w + 2 * pad - kw) / stride + 1;
  const int K = kh * kw * cin, M = batch * oh * ow;
  return (size_t)wgrad_shape(cout, K, M).S * cout * K * sizeof(float);
}

/* dW[cout][kh][kw][cin] (the packed layout of dana_conv2d_nhwc) (+)= sum over output pixels of
 * grad_out[m][cout] * input patch. grad_out is the gradient w.r.t. the conv output (already multiplied by the
 * frozen-BN scale and the ReLU mask by the caller); cin % 64 == 0, cout % 4 == 0. */
int dana_conv2d_wgrad_nhwc(const float* grad_out, const float* input, float* grad_weight, int batch, int in_h,
                           int in_w, int cin, int cout, int kh, int kw, int stride, int pad, long in_pix_stride,
                           long grad_pix_stride, const float* row_scale, int accumulate, void* workspace,
                           size_t workspace_bytes, dana_stream_t stream) {
  DANA_CHECK_ARG(batch >= 0 && in_h > 0 && in_w > 0 && cin > 0 && cout > 0 && kh > 0 && kw > 0 && stride > 0 && pad >= 0,
                 "dana_conv2d_wgrad_nhwc: bad shape");
  if (batch == 0) return DANA_OK;
  DANA_CHECK_ARG(grad_out && input && grad_weight, "dana_conv2d_wgrad_nhwc: null pointer");
  DANA_CHECK_ARG(cin % 64 == 0 && cout % 4 == 0, "dana_conv2d_wgrad_nhwc: needs cin %% 64 == 0 and cout %% 4 == 0");
  WgradParams p;
  memset(&p, 0, sizeof(p));
  p.dY = grad_out;
  p.X = input;
  p.IH = in_h;
  p.IW = in_w;
  p.OH = (in_h + 2 * pad - kh) / stride + 1;
  p.OW = (in_w + 2 * pad - kw) / stride + 1;
  DANA_CHECK_ARG(p.OH > 0 && p.OW > 0, "dana_conv2d_wgrad_nhwc: empty output");
  p.M = batch * p.OH * p.OW;
  p.N = cout;
  p.K = kh * kw * cin;
  p.Cin = cin;
  p.KH = kh;
  p.KW = kw;
  p.stride = stride;
  p.pad = pad;
  p.ldx = (int)(in_pix_stride > 0 ? in_pix_stride : cin);
  p.ldy = (int)(grad_pix_stride > 0 ? grad_pix_stride : cout);
  DANA_CHECK_ARG(p.ldx % 4 == 0 && p.ldy % 4 == 0 && ((uintptr_t)grad_out & 15) == 0 && ((uintptr_t)input & 15) == 0 &&
                     ((uintptr_t)grad_weight & 15) == 0,
                 "dana_conv2d_wgrad_nhwc: strides / pointers must be 16-byte aligned");
  const long xb = (long)batch * in_h * in_w * p.ldx * 4, yb = (long)p.M * p.ldy * 4;
  DANA_CHECK_ARG(xb < (long)OOB && yb < (long)OOB, "dana_conv2d_wgrad_nhwc: operand span >= 2 GiB; split the batch");
  p.x_bytes = (unsigned)xb;
  p.y_bytes = (unsigned)yb;
  const WgradShape ws = wgrad_shape(cout, p.K, p.M);
  const int tn = ws.tn, tk = ws.tk, S = ws.S;
  p.m_chunk = ((p.M + S - 1) / S + 31) / 32 * 32;
  p.nsplit = S;
  const size_t need = (size_t)S * cout * p.K * sizeof(float);
  if (!workspace || workspace_bytes < need) {
    dana_set_error("dana_conv2d_wgrad_nhwc: workspace %zu < %zu", workspace_bytes, need);
    return DANA_ERR_WORKSPACE;
  }
  p.partial = (float*)workspace;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(tn, tk, S);
  if (ws.tile == 128)
    launch_wgrad_128(p, grid, s);
  else
    wgrad_f32_kernel<<<grid, 256, 0, s>>>(p);
  DANA_CHECK_LAUNCH("dana_conv2d_wgrad_nhwc");
  const long n4 = (long)cout * p.K / 4;
  wgrad_reduce_kernel<<<dana_ceil_div(n4, 64), 256, 0, s>>>((const float4*)workspace, (float4*)grad_weight, row_scale,
                                                             n4, p.K / 4, S, accumulate);
  DANA_CHECK_LAUNCH("dana_conv2d_wgrad_nhwc(reduce)");
  return DANA_OK;
}

/* Weights for the data gradient of a stride-1 conv: out[cin][kh][kw][cout] = w[cout][KH-1-kh][KW-1-kw][cin] * scale[cout]
 * (scale = the frozen-BN scale folded into the incoming gradient; NULL = 1). dgrad is then
 * dana_conv2d_nhwc(grad_out, out, ..., cin<->cout swapped, same pad). */
int dana_conv2d_dgrad_weight(const float* w_packed, const float* scale, float* out, int cout, int cin, int kh, int kw,
                             dana_stream_t stream) {
  DANA_CHECK_ARG(w_packed && out && cout > 0 && cin > 0 && kh > 0 && kw > 0, "dana_conv2d_dgrad_weight: bad args");
  DANA_CHECK_ARG(kh * kw <= 65535, "dana_conv2d_dgrad_weight: too many taps");
  dim3 grid(dana_ceil_div(cin, 32), dana_ceil_div(cout, 32), kh * kw);
  dgrad_weight_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(w_packed, scale, out, cout, cin, kh, kw);
  DANA_CHECK_LAUNCH("dana_conv2d_dgrad_weight");
  return DANA_OK;
}

/* Data gradient of a strided 1x1 conv: scatter compact[batch][oh][ow][C] to out[batch][ih][iw][C] at (oh*stride,
 * ow*stride), zero elsewhere (resnet.py:71: the stride-2 1x1 convs of the Caffe bottleneck). */
int dana_upsample_scatter_nhwc(const float* compact, float* out, const float* mask_act, int batch, int oh, int ow,
                               int ih, int iw, int channels, int stride, dana_stream_t stream) {
  DANA_CHECK_ARG(batch >= 0 && oh > 0 && ow > 0 && ih >= (oh - 1) * stride + 1 && iw >= (ow - 1) * stride + 1 &&
                     channels % 4 == 0 && stride > 0,
                 "dana_upsample_scatter_nhwc: bad shape");
  if (batch == 0) return DANA_OK;
  DANA_CHECK_ARG(compact && out, "dana_upsample_scatter_nhwc: null pointer");
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(out, 0, (size_t)batch * ih * iw * channels * sizeof(float), s) != hipSuccess) {
    dana_set_error("dana_upsample_scatter_nhwc: memset failed");
    return DANA_ERR_HIP;
  }
  const long total = (long)batch * oh * ow * (channels / 4);
  upsample_scatter_kernel<<<dana_ceil_div(total, 256), 256, 0, s>>>((const float4*)compact, (float4*)out,
                                                                    (const float4*)mask_act, oh, ow, ih, iw,
                                                                    channels / 4, stride, total);
  DANA_CHECK_LAUNCH("dana_upsample_scatter_nhwc");
  return DANA_OK;
}

/* compact[b][oh][ow][channels] = x[b][oh*stride][ow*stride][0..channels) (x pixel stride in_pix_stride, 0 = channels): the
 * rows a strided 1x1 conv reads (resnet.py:71, the first block of layer2-4), so that its weight gradient is a plain-row
 * contraction (the software-pipelined kernel) and the block's two strided convs share one copy. */
int dana_downsample_gather_nhwc(const float* x, float* compact, int batch, int ih, int iw, int channels, int stride,
                                long in_pix_stride, dana_stream_t stream) {
  DANA_CHECK_ARG(batch >= 0 && ih > 0 && iw > 0 && channels > 0 && channels % 4 == 0 && stride > 0,
                 "dana_downsample_gather_nhwc: bad shape");
  if (batch == 0) return DANA_OK;
  DANA_CHECK_ARG(x && compact, "dana_downsample_gather_nhwc: null pointer");
  const long ldx = in_pix_stride > 0 ? in_pix_stride : channels;
  DANA_CHECK_ARG(ldx % 4 == 0 && ldx >= channels && (((uintptr_t)x | (uintptr_t)compact) & 15) == 0,
                 "dana_downsample_gather_nhwc: rows must be 16-byte aligned");
  const int oh = (ih - 1) / stride + 1, ow = (iw - 1) / stride + 1;
  const long total = (long)batch * oh * ow * (channels / 4);
  downsample_gather_kernel<<<dana_ceil_div(total, 256), 256, 0, (hipStream_t)stream>>>(x, (float4*)compact, oh, ow, ih, iw,
                                                                                        channels / 4, stride, ldx, total);
  DANA_CHECK_LAUNCH("dana_downsample_gather_nhwc");
  return DANA_OK;
}

}  // extern "C"
