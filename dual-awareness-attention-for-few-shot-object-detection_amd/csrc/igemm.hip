// fp32 implicit-GEMM convolution / GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32).
//
// This one kernel family carries every dense contraction of the DAnA forward path
// (SURVEY.md Appendix A): the Caffe-ResNet-50 trunk (lib/model/framework/resnet.py:66-146,
// stride on the first 1x1), the RPN 3x3 conv + heads (lib/model/rpn/rpn.py:28-36), layer4 on the
// pooled RoIs (dana.py:346,387-389) and, as 1x1 "convs", every nn.Linear / bmm of the BA + CISA
// attention (dana.py:124-147, 266-290).
//
//   C[m][n] = epi( alpha * sum_k A(m,k) * Bw[n][k] )            m = output pixel, n = out channel
//
// Data layout in HBM: activations NHWC (pixel-major, channel-minor; `lda` floats between pixels so
// a tensor can live inside a wider concat buffer), weights [N][K] with k = (kh, kw, cin) cin-minor.
// Both MFMA operands are therefore "rows with K contiguous": A rows are gathered pixels (implicit
// im2col), B rows are filters. Epilogue fuses frozen-BN scale/shift or bias, residual add and ReLU
// (resnet.py:84-100) and writes with row stride `ldc` so producers can write straight into concat
// buffers (dana.py:153-154 torch.cat eliminated).
//
// Kernel structure (wave64, 4 waves as 2x2; block BM x BN x 32; wave tile = 32x32 MFMA tiles):
//  * staging loads are BRANCH-FREE buffer loads (buffer_load_dwordx4 with a raw SRD): padding taps,
//    rows >= M, filters >= N and the K tail all become an out-of-range offset, which the hardware
//    returns as zeros. Per row the thread keeps one 32-bit byte offset and a 64-bit tap-validity
//    mask computed once; a K-step adds a wave-uniform tap delta. No exec-mask branches, no 64-bit
//    address math in the loop, so hipcc can interleave the loads with the MFMAs.
//  * LDS rows are padded to 36 dwords: the ds_read_b128 fragment reads (16-lane groups, distinct
//    rows) and the 8-lane ds_write_b128 staging writes are bank-conflict-free. Lane (i, h) holds
//    k = 8c+4h..8c+4h+3 of row i for BOTH operands, so MFMA step s multiplies the k-pair
//    {8c+s, 8c+4+s}; any pairing is valid because A and B agree.
//  * pipeline: global->register prefetch of tile t+1 is issued before the MFMAs of tile t and
//    written to the other LDS buffer after them; one barrier per K-step.
//  * epilogue: accumulators go through LDS (the staging buffers are dead by then) so that each
//    thread finishes 4 consecutive channels of one pixel: float4 scale/shift, float4 residual
//    load, float4 store -- whole 256/512-byte rows per wave instead of 4-byte scattered stores.
#include "common.h"
#include "../../include/dana_hip_debug.h"
#include <stdlib.h>
#include <atomic>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct IgemmParams {
  const float* A;
  const float* Bw;
  float* C;
  const float* scale;     // [N] or null
  const float* shift;     // [N] or null
  const float* residual;  // [M][ldr] or null
  const float* mask;      // [M][ldm] or null: out = act > 0 ? out : 0 (ReLU adjoint fused into a data-gradient conv)
  long ldm;
  int M, N, K;
  int IH, IW, OH, OW, Cin, KH, KW, stride, pad;
  // optional second geometry segment: rows m >= M0 are pixels of images IH1 x IW1 that start `pix1`
  // pixels into A (query and support images share every trunk launch: twice the tiles, half the tail)
  int M0, IH1, IW1, OH1, OW1, pix1;
  float* C1;              // segment-1 output (row m - M0), row stride ldc1
  const float* residual1;
  long ldc1, ldr1;
  int lda, ldb;           // floats
  long ldc, ldr;
  long batch_a, batch_b, batch_c;  // blockIdx.z strides (floats)
  unsigned a_bytes, b_bytes;       // addressable span of one batch slice (buffer descriptor range)
  float alpha;
  int relu;
  int vec_io;  // C / residual rows are 16-byte aligned -> float4 epilogue
  int vec_ss;  // scale / shift are 16-byte aligned (or null) -> one float4 load each
  int tiles_m, tiles_n;
  // optional SECOND K segment (split kernel only): after K0 = Cin channels of A, the K walk continues through K1 channels
  // of A2, a [batch][IH2][IW2] NHWC map sampled at (oh * stride2, ow * stride2) -- a bottleneck's 1x1 expand conv and
  // its (strided) 1x1 downsample conv as ONE contraction over the concatenated channels (resnet.py:84-100)
  const float* A2;
  int lda2, IH2, IW2, stride2, K1;
  int IH21, IW21, pix21;  // A2's geometry for rows of the second geometry segment (image size, first pixel)
  int bpre;               // Bw holds pre-split bf16 planes [ldb / 16][3][N][16] (dana_split_weight), ldb = K rounded up to 16
  // fused bottleneck tail (FUSE kernels only): the BM x 64 tile of this conv goes through LDS into a SECOND contraction,
  // a 1x1 conv 64 -> N2 with pre-split weights Bw2 [4][3][N2][16]; scale / shift / relu belong to the first conv,
  // scale2 / shift2 / residual / relu2 / C / ldc to the second
  const void* Bw2;
  const float* scale2;
  const float* shift2;
  int N2, relu2;
  unsigned a2_bytes;
  // A as pre-split planes (igemm_pp_kernel): A = bf16 planes [Kp / 16][3][a_rows][16] of the activation rows (the layout of
  // dana_split_weight with n = a_rows), written by the producing kernel's epilogue or by dana_split_weight; GEMM geometry only
  int apre, a_rows;
  unsigned long long* trace;  // debug (dana_set_igemm_trace): per block {start, first MFMA, loop end, end} in 100 MHz ticks + HW id
};

constexpr int BK = 32;
constexpr int LDS_LD = 36;             // padded staging row, dwords
constexpr unsigned OOB = 0x80000000u;  // > any descriptor range: the load returns zeros

// bijective XCD-aware remap: consecutive tile ids land on the same XCD's L2 (guide T1)
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg / 8, r = nwg % 8;
  const int xcd = bid % 8, local = bid / 8;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
}

__device__ __forceinline__ float4 ldg_b128(__amdgpu_buffer_rsrc_t r, unsigned off) {
  u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
  return *(float4*)&v;
}

template <int BM, int BN, int STEM>
__global__ void __launch_bounds__(256, 2) igemm_f32_kernel(IgemmParams p) {
  constexpr int TM = BM / 64, TN = BN / 64;  // 32x32 MFMA tiles per wave in m / n
  constexpr int RA = BM / 32, RB = BN / 32;  // float4 loads per thread per K-step
  constexpr int CLD = BN + 4;                // epilogue C-tile row, dwords
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                    // [2][BM][LDS_LD]
  float* Bs = smem + 2 * BM * LDS_LD;  // [2][BN][LDS_LD]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;

  // XCD-aware order over the WHOLE grid (batch slices included): the hardware deals linear workgroup ids (x fastest,
  // then z) round-robin to the 8 XCDs; the remap gives each XCD one contiguous run of (slice, tile) pairs, so that the
  // tiles of one batch slice -- one Winograd plane, one image's attention matrix -- share an L2 instead of each XCD
  // fetching its own copy of that slice's operands
  const int vid = xcd_remap((int)(blockIdx.z * gridDim.x + blockIdx.x), (int)(gridDim.x * gridDim.z));
  const int zb = vid / (int)gridDim.x;
  const int tile = vid - zb * (int)gridDim.x;
  const int tm_idx = tile / p.tiles_n, tn_idx = tile % p.tiles_n;
  const int m0 = tm_idx * BM, n0 = tn_idx * BN;
  const float* Ab = p.A + (long)zb * p.batch_a;
  const float* Bb = p.Bw + (long)zb * p.batch_b;
  float* Cb = p.C + (long)zb * p.batch_c;
  const __amdgpu_buffer_rsrc_t ra_src = __builtin_amdgcn_make_buffer_rsrc((void*)Ab, 0, (int)p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb_src = __builtin_amdgcn_make_buffer_rsrc((void*)Bb, 0, (int)p.b_bytes, 0x00020000);

  // ---- per-thread staging coordinates (computed once) -----------------------------------------
  const int c4 = tid & 7;   // which float4 of the 32-wide k chunk
  const int r0 = tid >> 3;  // row within a 32-row slab
  unsigned a_off[RA];       // byte offset of (pixel at tap (0,0), channel c4*4)
  unsigned long long a_mask[RA];  // bit t: tap t reads inside the image (and the row exists)
  bool a_seg1[RA];                // row belongs to the second geometry segment
#pragma unroll
  for (int j = 0; j < RA; ++j) {
    const int m = m0 + r0 + 32 * j;
    const bool ok = m < p.M;
    const bool s1 = ok && m >= p.M0;
    a_seg1[j] = s1;
    const int mm = ok ? (s1 ? m - p.M0 : m) : 0;
    const int IH = s1 ? p.IH1 : p.IH, IW = s1 ? p.IW1 : p.IW, OW = s1 ? p.OW1 : p.OW;
    const int ohw = (s1 ? p.OH1 : p.OH) * OW;
    const int img = mm / ohw, rem = mm - img * ohw;
    const int oh = rem / OW, ow = rem - oh * OW;
    const int ih0 = oh * p.stride - p.pad, iw0 = ow * p.stride - p.pad;
    const int pix = (s1 ? p.pix1 : 0) + (img * IH + ih0) * IW;
    unsigned long long mask = 0;
    if (STEM) {
      // chunk = filter row kh; this thread's float4 is tap kw = c4 (8th tap / 4th channel: zero weights)
      const int iw = iw0 + c4;
      if (ok && c4 < 7 && iw >= 0 && iw < IW)
        for (int kh = 0; kh < 7; ++kh)
          if (ih0 + kh >= 0 && ih0 + kh < IH) mask |= 1ull << kh;
      a_off[j] = (unsigned)((pix + iw) * 16);
    } else {
      if (ok)
        for (int kh = 0; kh < p.KH; ++kh)
          for (int kw = 0; kw < p.KW; ++kw)
            if (ih0 + kh >= 0 && ih0 + kh < IH && iw0 + kw >= 0 && iw0 + kw < IW)
              mask |= 1ull << (kh * p.KW + kw);
      a_off[j] = (unsigned)(((pix + iw0) * p.lda + c4 * 4) * 4);
    }
    a_mask[j] = mask;
  }
  unsigned b_off[RB];
#pragma unroll
  for (int j = 0; j < RB; ++j) {
    const int n = n0 + r0 + 32 * j;
    b_off[j] = n < p.N ? (unsigned)((n * p.ldb + c4 * 4) * 4) : OOB;
  }

  // two staging register sets: the loads of K-step kt+2 are issued before the MFMAs of K-step kt, so they
  // have two compute phases to land (and a 2-step conv has its whole K range in flight from the start)
  float4 ra0[RA], rb0[RB], ra1[RA], rb1[RB];
  auto load_tile = [&](int kt, float4(&ra)[RA], float4(&rb)[RB]) {
    const int k0 = kt * BK;
    const bool kok = (k0 + c4 * 4) < p.K;
    int tap;
    unsigned delta, delta1;  // wave-uniform tap offsets for segment 0 / 1 (they differ in image width only)
    if (STEM) {
      tap = kt;
      delta = (unsigned)(kt * p.IW * 16);
      delta1 = (unsigned)(kt * p.IW1 * 16);
    } else {
      tap = k0 / p.Cin;
      const int cin0 = k0 - tap * p.Cin;
      const int kh = tap / p.KW, kw = tap - kh * p.KW;
      delta = (unsigned)(((kh * p.IW + kw) * p.lda + cin0) * 4);
      delta1 = (unsigned)(((kh * p.IW1 + kw) * p.lda + cin0) * 4);
    }
#pragma unroll
    for (int j = 0; j < RA; ++j) {
      const bool ok = kok && ((a_mask[j] >> tap) & 1ull);
      ra[j] = ldg_b128(ra_src, ok ? a_off[j] + (a_seg1[j] ? delta1 : delta) : OOB);
    }
#pragma unroll
    for (int j = 0; j < RB; ++j) rb[j] = ldg_b128(rb_src, (kok && b_off[j] != OOB) ? b_off[j] + k0 * 4 : OOB);
  };
  auto store_tile = [&](int buf, const float4(&ra)[RA], const float4(&rb)[RB]) {
    float* as = As + buf * BM * LDS_LD;
    float* bs = Bs + buf * BN * LDS_LD;
#pragma unroll
    for (int j = 0; j < RA; ++j) *(float4*)(as + (r0 + 32 * j) * LDS_LD + c4 * 4) = ra[j];
#pragma unroll
    for (int j = 0; j < RB; ++j) *(float4*)(bs + (r0 + 32 * j) * LDS_LD + c4 * 4) = rb[j];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // Residual prefetch (64x64 tiles): the 1x1 expand convs of a bottleneck have K = 64..512, i.e. 2-16
  // K-steps, and are HBM-bound on their [M][N] residual + output. Issuing the residual tile's loads
  // here puts them in flight under the whole K loop instead of serialising them in the epilogue.
  constexpr int TPR = BN / 4;     // threads per output row (epilogue mapping)
  constexpr int RPP = 256 / TPR;  // rows per pass
  constexpr bool PREFETCH_RES = (BM / RPP) <= 4;
  const int ec = (tid % TPR) * 4;  // this thread's 4 columns within the tile
  const int er = tid / TPR;
  float4 rpre[PREFETCH_RES ? BM / RPP : 1];
  if (PREFETCH_RES && p.residual && p.vec_io) {
#pragma unroll
    for (int q = 0; q < BM / RPP; ++q) {
      const int m = m0 + er + q * RPP;
      const int n = n0 + ec;
      rpre[q] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m < p.M && n + 3 < p.N) {
        const bool s1 = m >= p.M0;
        rpre[q] = *(const float4*)(s1 ? p.residual1 + (long)(m - p.M0) * p.ldr1 + n : p.residual + (long)m * p.ldr + n);
      }
    }
  }

  const int nk = (p.K + BK - 1) / BK;
  load_tile(0, ra0, rb0);
  if (nk > 1) load_tile(1, ra1, rb1);
  store_tile(0, ra0, rb0);
  __syncthreads();

  auto k_step = [&](int kt, float4(&rfree_a)[RA], float4(&rfree_b)[RB], const float4(&rnext_a)[RA],
                    const float4(&rnext_b)[RB]) {
    const int buf = kt & 1;
    if (kt + 2 < nk) load_tile(kt + 2, rfree_a, rfree_b);  // rfree held K-step kt, already staged in LDS
    const float* as = As + buf * BM * LDS_LD + (wm * (BM / 2) + li) * LDS_LD + lh * 4;
    const float* bs = Bs + buf * BN * LDS_LD + (wn * (BN / 2) + li) * LDS_LD + lh * 4;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float4 af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = *(const float4*)(as + i * 32 * LDS_LD + c * 8);
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = *(const float4*)(bs + j * 32 * LDS_LD + c * 8);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
        }
    }
    if (kt + 1 < nk) store_tile(buf ^ 1, rnext_a, rnext_b);  // K-step kt+1, issued one step ago
    __syncthreads();
  };
  for (int kt = 0; kt < nk; kt += 2) {
    k_step(kt, ra0, rb0, ra1, rb1);
    if (kt + 1 < nk) k_step(kt + 1, ra1, rb1, ra0, rb0);
  }

  // ---- epilogue through LDS: C/D map col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5) -------
  float* Cs = smem;  // [BM][CLD]; staging buffers are dead after the loop's last barrier
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      float* cw = Cs + (wm * (BM / 2) + i * 32 + 4 * lh) * CLD + wn * (BN / 2) + j * 32 + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) cw[((r & 3) + 8 * (r >> 2)) * CLD] = acc[i][j][r];
    }
  __syncthreads();
  const int n = n0 + ec;
  float sc[4], sh[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const bool nok = (n + q) < p.N;
    sc[q] = (nok && p.scale) ? p.scale[n + q] : 1.f;
    sh[q] = (nok && p.shift) ? p.shift[n + q] : 0.f;
  }
  const bool full4 = p.vec_io && (n + 3) < p.N;
#pragma unroll
  for (int q = 0; q < BM / RPP; ++q) {
    const int rr = er + q * RPP;
    const int m = m0 + rr;
    if (m >= p.M) break;
    const float4 a4 = *(const float4*)(Cs + rr * CLD + ec);
    float v[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = v[q] * p.alpha * sc[q] + sh[q];
    const bool s1 = m >= p.M0;
    float* cp = s1 ? p.C1 + (long)(m - p.M0) * p.ldc1 + n : Cb + (long)m * p.ldc + n;
    const float* rp = s1 ? p.residual1 + (long)(m - p.M0) * p.ldr1 + n : p.residual + (long)m * p.ldr + n;
    if (full4) {
      if (p.residual) {
        const float4 r4 = PREFETCH_RES ? rpre[q] : *(const float4*)rp;
        v[0] += r4.x;
        v[1] += r4.y;
        v[2] += r4.z;
        v[3] += r4.w;
      }
      if (p.relu) {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.f);
      }
      if (p.mask) {
        const float4 k4 = *(const float4*)(p.mask + (long)m * p.ldm + n);
        v[0] = k4.x > 0.f ? v[0] : 0.f;
        v[1] = k4.y > 0.f ? v[1] : 0.f;
        v[2] = k4.z > 0.f ? v[2] : 0.f;
        v[3] = k4.w > 0.f ? v[3] : 0.f;
      }
      *(float4*)cp = make_float4(v[0], v[1], v[2], v[3]);
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if ((n + q) < p.N) {
          float x = v[q];
          if (p.residual) x += rp[q];
          if (p.relu) x = fmaxf(x, 0.f);
          if (p.mask && !(p.mask[(long)m * p.ldm + n + q] > 0.f)) x = 0.f;
          cp[q] = x;
        }
    }
  }
}

// ---- fp32 contraction on the bf16 matrix cores: exact three-way split, six products ------------------------------
// gfx950 runs v_mfma_f32_32x32x16_bf16 at 16x the rate of v_mfma_f32_32x32x2_f32. An fp32 value x is EXACTLY
// x = h + m + l with h, m, l three bf16 numbers (8 significant bits each, taken by truncation: h = top 16 bits of x,
// m = top 16 bits of x - h, l = x - h - m), so a*b = sum of nine bf16 x bf16 products, each of which the matrix core
// forms exactly and accumulates in fp32. The six products of weight >= 2^-16 are issued (hh, hm, mh, mm, hl, lh); the
// three dropped ones are <= 2^-23 |a||b| together, i.e. below the rounding of the fp32 accumulation itself -- operands,
// accumulation and results are fp32, only the multiplier array is the bf16 one (tests/test_gpu_contractions.py checks
// both kernels against an fp64 contraction: same error level). 6 x 32 cycles per K = 16 instead of 8 x 64: 2.67x
// the f32-MFMA ceiling.
//
// Structure: block BM x BN x 16, 4 waves as 2x2, wave tile (BM/2) x (BN/2). The staging path is the f32 kernel's
// (branch-free buffer loads of fp32 rows, two register sets in flight); the split happens once per element on the
// way into LDS (4 VALU + 1.5 v_perm per element), which holds three bf16 planes per operand: rows of 16 bf16 padded
// to 48 bytes (ds_write_b64 staging writes and ds_read_b128 fragment reads are bank-conflict-free). A lane's b128
// fragment is k = 8*(lane>>5) .. +7 of row lane&31 -- the operand layout of the 32x32x16 MFMA.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int SBK = 16;  // K per step
// LDS row of one plane: 16 bf16 = 32 B, UNPADDED (round 3; rounds 1-2 padded to 48 B). The two 16-byte halves of row r
// (k 0..7, k 8..15) swap places when bit 3 of r is set: with that XOR the ds_read_b128 fragment reads (hardware lane groups
// {0-3,12-15,20-27}, ...: 16 rows each) and the ds_write_b64 / b128 staging writes (16 / 8 consecutive lanes = 128
// consecutive bytes) are bank-conflict-free without padding, and the staging buffers of a 128x64 tile shrink from 55.3
// to 36.9 KB -- three workgroups per CU instead of two for the N <= 64 launches (160 registers).
constexpr int SLD = 8;
__device__ __forceinline__ int swz(int row) { return (row >> 3) & 1; }

template <int V>
struct IC {
  static constexpr int value = V;
};
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(IC<I>{});
    static_for<I + 1, N>(f);
  }
}

__device__ __forceinline__ void split3(const float4& v, uint2& h, uint2& m, uint2& l) { dana_split3(v, h, m, l); }

// Issue schedule of one K-step of the split kernel. Everything that is not an MFMA is cut into micro-items of one or two
// INDEPENDENT instructions and dealt out over the NM gaps between the step's MFMAs so that every gap carries about the
// same number of issue slots (the 8-pass MFMA hides about five; a wave alone on its SIMD -- a launch's last round, a
// tile-starved layer -- has nothing else to cover a gap that carries more):
//   kind 0  one half of a global load: [compare + select] / [load + offset step]            2 x NL items, weight 2
//   kind 1  one fragment read (ds_read_b128) of the NEXT step's operands                    NR items, weight 1
//   kind 2  one thirteenth of a float4's three-way split: H01 H23 R01 R23 M01 M23 L01 L23 (two elements each, so
//           that consecutive instructions never depend on each other), PA PB PC (the packs), W1 W2 (staging writes)
//   kind 3  one staging write of a PRE-SPLIT filter chunk (ds_write_b128 of 8 bf16 as loaded)   NBW items, weight 1
// Every kind has its own time line over the step: the loads go into its first third (their data is used one step
// later), the reads and the split are spread evenly; an item's gap is its time scaled to NM.
template <int NM, int NL, int NR, int NCVF, int NBW>
struct StepSched {
  static constexpr int NCV = 13 * NCVF, NLH = 2 * NL, NIT = NCV + NR + NLH + NBW;
  // The step's ONE barrier sits behind MFMA number SB, not at the step boundary: it orders the LDS traffic of step t
  // (staging writes of tile t+2, fragment reads of tile t+1) against the LDS traffic of step t+1, and the MFMAs of
  // step t+1 read registers only. So the first SB MFMAs of a step run while the previous step's last staging writes
  // drain, every wave reaches the barrier with lgkmcnt already at zero, and the matrix pipe does not idle through an
  // LDS round trip per step (with one wave per SIMD that was a third of the step: tools/ubench/split_loop.hip).
  // All LDS items of a step are placed in gaps >= SB.
  static constexpr int SB = NM >= 24 ? 4 : (NM >= 12 ? 3 : 1);
  int kind[NIT], idx[NIT], gap[NIT];
};
template <int NM, int NL, int NR, int NCVF, int NBW>
constexpr StepSched<NM, NL, NR, NCVF, NBW> make_step_sched() {
  using S = StepSched<NM, NL, NR, NCVF, NBW>;
  S s{};
  int n = 0;
  // times in 1/10000 of a step. With pre-split filters the split work is half: it starts a quarter into the step, so
  // that the activation rows requested at the start of the previous step have a step and a quarter to land
  const long cv0 = NBW ? 2500 : 0;
  auto put = [&](int kind, int idx, long t, bool lds) {
    int g = (int)(t * NM / 10000);
    g = g < NM ? g : NM - 1;
    if (lds && g < S::SB) g = S::SB;
    s.kind[n] = kind;
    s.idx[n] = idx;
    s.gap[n++] = g;
  };
  // (items of one gap are emitted in this order: loads, pre-split writes, reads, split)
  for (int l = 0; l < S::NLH; ++l) put(0, l, (2 * l + 1) * 3000 / (2 * S::NLH), false);
  for (int b = 0; b < NBW; ++b) put(3, b, 1800 + (2 * b + 1) * 800 / (2 * (NBW ? NBW : 1)), true);
  for (int r = 0; r < NR; ++r) put(1, r, 1700 + (2 * r + 1) * 8000 / (2 * NR), true);
  for (int i = 0; i < S::NCV; ++i) put(2, i, cv0 + (2 * i + 1) * (9950 - cv0) / (2 * S::NCV), i % 13 >= 11);
  return s;
}
template <int NM, int NL, int NR, int NCVF, int NBW>
inline constexpr StepSched<NM, NL, NR, NCVF, NBW> kStepSched = make_step_sched<NM, NL, NR, NCVF, NBW>();

// ---- epilogue on the accumulator registers (C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)).
// With the output channel on the lane axis one accumulator register across a half-wave is 32 consecutive floats of one
// output row = one 128-byte segment, so scale / shift / residual / ReLU / mask run where the values are and every access is
// a dword buffer access (descriptor + 32-bit lane offset + a uniform row term): no LDS C tile (49 instead of 67.6 KB per
// 128 x 128 tile), no barrier, no alignment condition on the rows. Rows past M fall outside the descriptors' ranges (loads
// return zeros, stores are dropped); lanes whose channel is past N carry an out-of-range offset. Same arithmetic, in the
// same order, as the LDS form of igemm_split_body (ELDS: dana_set_epilogue_mode(1)) -> the same bits.
template <int BM, int BN, int HALVES = 0, int WITH_MASK = 1>
__device__ __forceinline__ void epilogue_regs(const IgemmParams& p, f32x16 (&acc)[BM / 64][BN / 64], int m0, int n0, float* Cb,
                                              int wm, int wn, int li, int lh) {
  constexpr int TM = BM / 64, TN = BN / 64;
  // (uniform) the tile lies in ONE geometry segment -- rows past M are not a segment: the descriptors' ranges drop them
  const bool one_seg = m0 + BM <= p.M0 || m0 >= p.M0 || p.M0 >= p.M;
  float scv[TN], shv[TN];
  int ncol[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    ncol[j] = n0 + wn * (BN / 2) + j * 32 + li;
    const bool nok = ncol[j] < p.N;
    scv[j] = ((nok && p.scale) ? p.scale[ncol[j]] : 1.f) * p.alpha;
    shv[j] = (nok && p.shift) ? p.shift[ncol[j]] : 0.f;
  }
  if (one_seg) {
    const bool seg1 = m0 >= p.M0;
    const long ld_c = seg1 ? p.ldc1 : p.ldc, ld_r = seg1 ? p.ldr1 : p.ldr;
    const long mrel = seg1 ? m0 - p.M0 : m0;
    const int rows_valid = (p.M - m0) < BM ? (p.M - m0) : BM;
    const int ldc4 = (int)ld_c * 4, ldr4 = (int)ld_r * 4, ldm4 = (int)p.ldm * 4;
    const __amdgpu_buffer_rsrc_t rc_dst = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((seg1 ? p.C1 : Cb) + mrel * ld_c), 0, rows_valid * ldc4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rc_res = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.residual ? (seg1 ? p.residual1 : p.residual) + mrel * ld_r : p.A), 0, p.residual ? rows_valid * ldr4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rc_msk = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.mask ? p.mask + (long)m0 * p.ldm : p.A), 0, p.mask ? rows_valid * ldm4 : 0, 0x00020000);
    const int lrow = wm * (BM / 2) + 4 * lh;  // the lane's first row of the tile
    unsigned vo_c[TN], vo_r[TN], vo_m[TN];    // lane offsets (bytes) at accumulator row 0; out of range past N
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const bool nok = ncol[j] < p.N;
      vo_c[j] = nok ? (unsigned)(lrow * ldc4 + ncol[j] * 4) : OOB;
      vo_r[j] = nok ? (unsigned)(lrow * ldr4 + ncol[j] * 4) : OOB;
      vo_m[j] = nok ? (unsigned)(lrow * ldm4 + ncol[j] * 4) : OOB;
    }
    auto run = [&](auto res_c, auto relu_c, auto mask_c) {
      constexpr bool RES = decltype(res_c)::value != 0, RELU = decltype(relu_c)::value != 0, MASK = decltype(mask_c)::value != 0;
      // Without a mask all residual rows of the tile are requested before the first is used (one memory round trip per
      // tile); a data-gradient launch (mask rows on top) works in halves of 32 rows per wave so that the requested rows
      // stay inside the two-workgroups-per-CU register budget.
      constexpr int HI = (MASK || HALVES) ? 1 : TM;  // accumulator tile rows (i) per pass (HALVES: a kernel on a three-
                                                     // workgroups-per-CU register budget requests 32 residual values at a time)
#pragma unroll
      for (int i0 = 0; i0 < TM; i0 += HI) {
        float res[RES ? HI : 1][TN][16], msk[MASK ? HI : 1][TN][16];
#pragma unroll
        for (int ii = 0; ii < HI; ++ii)
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int rt = (i0 + ii) * 32 + (r & 3) + 8 * (r >> 2);
              if constexpr (RES)
                res[ii][j][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rc_res, (int)(vo_r[j] + (unsigned)(rt * ldr4)), 0, 0));
              if constexpr (MASK)
                msk[ii][j][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rc_msk, (int)(vo_m[j] + (unsigned)(rt * ldm4)), 0, 0));
            }
#pragma unroll
        for (int ii = 0; ii < HI; ++ii)
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int rt = (i0 + ii) * 32 + (r & 3) + 8 * (r >> 2);
              float v = acc[i0 + ii][j][r] * scv[j] + shv[j];
              if constexpr (RES) v += res[ii][j][r];
              if constexpr (RELU) v = fmaxf(v, 0.f);
              if constexpr (MASK) v = msk[ii][j][r] > 0.f ? v : 0.f;
              __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rc_dst, (int)(vo_c[j] + (unsigned)(rt * ldc4)), 0, 0);
            }
      }
    };
    if (WITH_MASK && p.mask) {  // (data gradients)
      if (p.residual) {
        if (p.relu) run(IC<1>{}, IC<1>{}, IC<1>{});
        else run(IC<1>{}, IC<0>{}, IC<1>{});
      } else {
        if (p.relu) run(IC<0>{}, IC<1>{}, IC<1>{});
        else run(IC<0>{}, IC<0>{}, IC<1>{});
      }
    } else if (p.residual) {
      if (p.relu) run(IC<1>{}, IC<1>{}, IC<0>{});
      else run(IC<1>{}, IC<0>{}, IC<0>{});
    } else {
      if (p.relu) run(IC<0>{}, IC<1>{}, IC<0>{});
      else run(IC<0>{}, IC<0>{}, IC<0>{});
    }
  } else {
    // the one tile row that straddles M0: every row picks its segment's output / residual rows (64-bit addresses)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (m < p.M && ncol[j] < p.N) {
            const bool s1 = m >= p.M0;
            float v = acc[i][j][r] * scv[j] + shv[j];
            if (p.residual) v += s1 ? p.residual1[(long)(m - p.M0) * p.ldr1 + ncol[j]] : p.residual[(long)m * p.ldr + ncol[j]];
            if (p.relu) v = fmaxf(v, 0.f);
            if (p.mask && !(p.mask[(long)m * p.ldm + ncol[j]] > 0.f)) v = 0.f;
            (s1 ? p.C1 + (long)(m - p.M0) * p.ldc1 : Cb + (long)m * p.ldc)[ncol[j]] = v;
          }
        }
  }
}

// BPRE: the B operand is a WEIGHT that was split once per weight version (dana_split_weight: three bf16 planes per
// K-step, [Kp / 16][3][N][16], Kp = K rounded up to 16, zero padded): its chunks of 8 bf16 go from HBM to the LDS planes as loaded, and
// the K loop splits the activation rows only -- half the VALU work of the step.
template <int BM, int BN, int STEM, int BPRE = 0, int FUSE = 0, int ELDS = 0>
__device__ __forceinline__ void igemm_split_body(const IgemmParams& p) {
  constexpr int TM = BM / 64, TN = BN / 64;  // 32x32 MFMA tiles per wave in m / n
  constexpr int RA = BM / 64;                // float4 loads of A per thread per K-step (64 rows per pass)
  // B: fp32 rows like A (64 rows per pass), or 16-byte chunks of the pre-split planes (3 planes x BN rows x 2 halves)
  constexpr int RB = BPRE ? (3 * BN * 2 + 255) / 256 : BN / 64;
  constexpr int CLD = BN + 4;                // epilogue C-tile row, dwords
  extern __shared__ __attribute__((aligned(16))) float smem[];
  unsigned* As = (unsigned*)smem;      // [2][3][BM][SLD]
  unsigned* Bs = As + 2 * 3 * BM * SLD;  // [2][3][BN][SLD]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;

  const unsigned long long t_start = p.trace ? __builtin_readcyclecounter() : 0ull;
  const unsigned long long w_start = p.trace ? wall_clock64() : 0ull;
  // XCD-aware order over the WHOLE grid (batch slices included): the hardware deals linear workgroup ids (x fastest,
  // then z) round-robin to the 8 XCDs; the remap gives each XCD one contiguous run of (slice, tile) pairs, so that the
  // tiles of one batch slice -- one Winograd plane, one image's attention matrix -- share an L2 instead of each XCD
  // fetching its own copy of that slice's operands
  const int vid = xcd_remap((int)(blockIdx.z * gridDim.x + blockIdx.x), (int)(gridDim.x * gridDim.z));
  const int zb = vid / (int)gridDim.x;
  const int tile = vid - zb * (int)gridDim.x;
  const int tm_idx = tile / p.tiles_n, tn_idx = tile % p.tiles_n;
  const int m0 = tm_idx * BM, n0 = tn_idx * BN;
  const float* Ab = p.A + (long)zb * p.batch_a;
  const float* Bb = p.Bw + (long)zb * p.batch_b;
  float* Cb = p.C + (long)zb * p.batch_c;
  __amdgpu_buffer_rsrc_t ra_src = __builtin_amdgcn_make_buffer_rsrc((void*)Ab, 0, (int)p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb_src = __builtin_amdgcn_make_buffer_rsrc((void*)Bb, 0, (int)p.b_bytes, 0x00020000);

  // The K walk is over "taps" of `cin_t` channels. Conv / GEMM: tap = filter tap (kh, kw), cin_t = Cin, and this
  // thread's float4 is channels c4*4.. of the pixel. 7x7 stem over NHWC4 (weights [cout][7][8][4], K = 224): a tap is
  // half a filter row -- four pixels x four channels = 16 k -- i.e. (kh, half), and the float4 is pixel kw = 4*half+c4.
  int cin_t = STEM ? SBK : p.Cin;
  const int kw_t = STEM ? 2 : p.KW;                // taps per filter row
  const int kh_t = STEM ? 7 : p.KH;
  const int lda4 = (STEM ? 4 : p.lda) * 4;          // bytes between pixels
  const int kwstep = STEM ? 4 * lda4 : lda4;        // bytes between taps of a row
  const int c4 = tid & 3;   // which float4 of the 16-wide k chunk
  const int r0 = tid >> 2;  // row within a 64-row slab
  // LDS column (dwords) of this lane's staging write / fragment read inside a 32-byte row (see SLD: swizzled halves)
  const int wcol = ((c4 >> 1) ^ swz(r0)) * 4 + (c4 & 1) * 2;
  const int rcol = (lh ^ swz(li)) * 4;
  const int klim = p.K - c4 * 4;  // this lane's float4 of a K-step starting at k0 is inside K iff k0 < klim
  constexpr int DEAD = (int)0x80000000;  // k0 < DEAD never holds
  unsigned a_off[RA];   // byte offset of this lane's float4 at tap (0, 0), channel chunk 0
  unsigned a_mask[RA];  // bit t: tap t reads inside the image (and the row exists); <= 32 taps on this kernel
  int a_dseg[RA];       // bytes added per filter row on top of segment 0's row pitch (second geometry segment)
  unsigned a_off2[RA];  // second K segment (A2): byte offset of this lane's float4 of the row's pixel, or OOB
  // 1x1 / stride 1 / no padding (every GEMM, most convs of the path): output row m IS input pixel m -- no divisions
  const bool lin = !STEM && p.KH * p.KW == 1 && p.stride == 1 && p.pad == 0;
#pragma unroll
  for (int j = 0; j < RA; ++j) {
    const int m = m0 + r0 + 64 * j;
    const bool ok = m < p.M;
    const bool s1 = ok && m >= p.M0;
    const int mm = ok ? (s1 ? m - p.M0 : m) : 0;
    if (lin && !p.A2) {
      a_off[j] = (unsigned)(((s1 ? p.pix1 : 0) + mm) * lda4 + c4 * 16);
      a_mask[j] = ok ? 1u : 0u;
      a_dseg[j] = 0;
      a_off2[j] = OOB;
      continue;
    }
    const int IH = s1 ? p.IH1 : p.IH, IW = s1 ? p.IW1 : p.IW, OW = s1 ? p.OW1 : p.OW;
    const int ohw = (s1 ? p.OH1 : p.OH) * OW;
    const int img = mm / ohw, rem = mm - img * ohw;
    const int oh = rem / OW, ow = rem - oh * OW;
    const int ih0 = oh * p.stride - p.pad, iw0 = ow * p.stride - p.pad;
    const int pix = (s1 ? p.pix1 : 0) + (img * IH + ih0) * IW;
    unsigned mask = 0;
    if (STEM) {
      // tap t = 2 * kh + half is inside the image iff row ih0 + kh is and this lane's pixel iw0 + 4 * half + c4 is (and is
      // one of the 7 filter columns): the row range as a run of bit pairs, the two halves as a 2-bit pattern
      const int lo = ih0 < 0 ? -ih0 : 0, hi = IH - ih0 < 7 ? IH - ih0 : 7;
      const int iwa = iw0 + c4, iwb = iw0 + 4 + c4;
      const unsigned cv = ((iwa >= 0 && iwa < IW) ? 0x1555u : 0u) | ((iwb >= 0 && iwb < IW && c4 < 3) ? 0x2aaau : 0u);
      if (ok && hi > lo) mask = ((1u << (2 * hi)) - (1u << (2 * lo))) & cv;
    } else if (ok) {
      unsigned colbits = 0;
      for (int kw = 0; kw < kw_t; ++kw)
        if (iw0 + kw >= 0 && iw0 + kw < IW) colbits |= 1u << kw;
      for (int kh = 0; kh < kh_t; ++kh)
        if (ih0 + kh >= 0 && ih0 + kh < IH) mask |= colbits << (kh * kw_t);
    }
    a_off[j] = STEM ? (unsigned)((pix + iw0 + c4) * lda4) : (unsigned)((pix + iw0) * lda4 + c4 * 16);
    a_mask[j] = mask;
    a_dseg[j] = s1 ? (p.IW1 - p.IW) * lda4 : 0;
    a_off2[j] = (!STEM && p.A2 && ok)
                    ? (unsigned)(((s1 ? p.pix21 : 0) + ((long)img * (s1 ? p.IH21 : p.IH2) + oh * p.stride2) * (s1 ? p.IW21 : p.IW2) +
                                  ow * p.stride2) * p.lda2 * 4 + c4 * 16)
                    : OOB;
  }
  unsigned b_cur[RB];  // byte offset of this lane's float4 of the current K-step in its filter row
  int b_lim[RB];
  int b_lds[RB];       // (BPRE) dword offset of this lane's chunk inside one staging buffer of B
#pragma unroll
  for (int j = 0; j < RB; ++j) {
    if constexpr (BPRE) {
      // chunk q = (plane, row, half): 8 bf16 = k 8*half.. of one filter row of one plane; p.ldb = Kp
      // (a chunk index past the three planes -- 64-row tiles: 384 chunks on 512 slots -- repeats the lane's first
      // chunk: same load, same LDS bytes, no branch in the stream)
      const int q = (tid + 256 * j < 6 * BN) ? tid + 256 * j : tid;
      const int pl = q / (2 * BN), rem = q - pl * (2 * BN);
      const int row = rem >> 1, half = rem & 1;
      const int n = n0 + row;
      b_cur[j] = (unsigned)((((long)pl * p.N + n) * SBK + half * 8) * 2);  // (K-step 0; a K-step is 3 * N * 32 bytes)
      b_lim[j] = n < p.N ? (p.K + SBK - 1) / SBK * SBK : DEAD;  // (rows past N: zeros; the planes are zero padded to 16 k)
      b_lds[j] = (pl * BN + row) * SLD + (half ^ swz(row)) * 4;
    } else {
      const int n = n0 + r0 + 64 * j;
      b_cur[j] = (unsigned)((n * p.ldb + c4 * 4) * 4);
      b_lim[j] = n < p.N ? klim : DEAD;
      b_lds[j] = 0;
    }
  }
  const unsigned B_STEP = BPRE ? (unsigned)p.N * 3u * SBK * 2u : SBK * 4;  // bytes from one K-step's chunk to the next

  float4 ra0[RA], rb0[RB], ra1[RA], rb1[RB];
  // K-steps are loaded in order: per step and load one compare (k0 against a per-lane limit that is DEAD for padding
  // taps, missing rows / filters and the K tail), one select (offset or out-of-range -> the load returns zeros) and
  // one add; the per-tap state is rebuilt under a scalar branch when a tap's channels run out (every Cin/16 steps).
  // A step past the end of K reads nothing, which keeps the main loop free of tail conditions.
  int lt_k0 = 0, lt_cin = 0, lt_kh = 0, lt_kw = 0;
  unsigned lt_bit = 1u;
  unsigned a_cur[RA];
  int a_lim[RA];
#pragma unroll
  for (int j = 0; j < RA; ++j) {
    a_cur[j] = a_off[j];
    a_lim[j] = (a_mask[j] & 1u) ? klim : DEAD;
  }
  auto next_tap = [&]() {  // rare: the tap's channels are used up
    if (lt_cin >= cin_t) {
      lt_cin = 0;
      if (!STEM && p.A2) {
        // the second K segment: another tensor (own descriptor, own pixel map), K1 channels; nothing after it
        ra_src = __builtin_amdgcn_make_buffer_rsrc((void*)p.A2, 0, (int)p.a2_bytes, 0x00020000);
        cin_t = 0x7fffffff;  // (steps past K read nothing: k0 >= klim)
#pragma unroll
        for (int j = 0; j < RA; ++j) {
          a_cur[j] = a_off2[j];
          a_lim[j] = a_off2[j] != OOB ? klim : DEAD;
        }
        return;
      }
      lt_bit <<= 1;
      if (++lt_kw == kw_t) {
        lt_kw = 0;
        ++lt_kh;
      }
      const int d0 = lt_kh * p.IW * lda4 + lt_kw * kwstep;
#pragma unroll
      for (int j = 0; j < RA; ++j) {
        a_cur[j] = a_off[j] + (unsigned)(d0 + lt_kh * a_dseg[j]);
        a_lim[j] = (a_mask[j] & lt_bit) ? klim : DEAD;
      }
    }
  };
  auto load_a = [&](int j, float4& dst) {
    unsigned off = lt_k0 < a_lim[j] ? a_cur[j] : OOB;
    asm volatile("" : "+v"(off));
    dst = ldg_b128(ra_src, off);
    a_cur[j] += SBK * 4;
  };
  auto load_b = [&](int j, float4& dst) {
    unsigned off = lt_k0 < b_lim[j] ? b_cur[j] : OOB;
    asm volatile("" : "+v"(off));
    dst = ldg_b128(rb_src, off);
    b_cur[j] += B_STEP;
  };
  auto load_tile = [&](float4(&ra)[RA], float4(&rb)[RB]) {
    next_tap();
#pragma unroll
    for (int j = 0; j < RB; ++j) load_b(j, rb[j]);
#pragma unroll
    for (int j = 0; j < RA; ++j) load_a(j, ra[j]);
    lt_k0 += SBK;
    lt_cin += SBK;
  };
  auto store_tile = [&](int buf, const float4(&ra)[RA], const float4(&rb)[RB]) {
    unsigned* as = As + buf * 3 * BM * SLD + r0 * SLD + wcol;
    unsigned* bs = Bs + buf * 3 * BN * SLD + r0 * SLD + wcol;
#pragma unroll
    for (int j = 0; j < RA; ++j) {
      uint2 h, m, l;
      split3(ra[j], h, m, l);
      *(uint2*)(as + (0 * BM + 64 * j) * SLD) = h;
      *(uint2*)(as + (1 * BM + 64 * j) * SLD) = m;
      *(uint2*)(as + (2 * BM + 64 * j) * SLD) = l;
    }
#pragma unroll
    for (int j = 0; j < RB; ++j) {
      if constexpr (BPRE) {
        *(float4*)(Bs + buf * 3 * BN * SLD + b_lds[j]) = rb[j];
      } else {
        uint2 h, m, l;
        split3(rb[j], h, m, l);
        *(uint2*)(bs + (0 * BN + 64 * j) * SLD) = h;
        *(uint2*)(bs + (1 * BN + 64 * j) * SLD) = m;
        *(uint2*)(bs + (2 * BN + 64 * j) * SLD) = l;
      }
    }
  };

  f32x16 acc[TM][TN];  // (zeroed below, under the first tiles' memory round trip)

  // ---- main loop: three stages in flight -------------------------------------------------------------------------
  //   registers R[.]  : fp32 rows of tile t+2 / t+3 (global loads, issued one iteration before they are split)
  //   LDS[.]          : bf16 planes of tile t+1 (written during iteration t-1) and, being written, tile t+2
  //   fragments F[.]  : tile t (read from LDS during iteration t-1) feeding this iteration's MFMAs, tile t+1 being read
  // One barrier per K-step and nothing of the step's own data movement in front of its MFMAs: with one wave per SIMD
  // (few tiles) the matrix pipe would otherwise idle through every address computation and LDS round trip.
  // The issue order is fixed by hand (sched_barrier fences + empty asm pins): the 8-pass MFMA occupies the matrix pipe
  // for 32 cycles and about five independent issues fit in its shadow, so all other work of the step is cut into items
  // -- address set-up, one global load, four fragment reads, one element's h/m/l (4 VALU), one pair's three packs (3
  // v_perm), one float4's three staging writes -- and dealt out evenly between the MFMAs.
  constexpr int NM = 6 * TM * TN;  // MFMAs per K-step
  constexpr int NF = RA + RB;      // 16-byte loads per thread per K-step
  constexpr int NCVF = BPRE ? RA : RA + RB;  // float4s split per thread per K-step
  constexpr int NBW = BPRE ? RB : 0;         // pre-split chunks written per thread per K-step
  constexpr int NR = 3 * (TM + TN);  // fragment reads per K-step
  using SCH = StepSched<NM, NF, NR, NCVF, NBW>;  // the step's issue schedule (make_step_sched)
  const int nk = (p.K + SBK - 1) / SBK;

  u32x4 fa0[3][TM], fb0[3][TN], fa1[3][TM], fb1[3][TN];  // fragments as raw dwords (8 bf16 each)
  auto read_frags = [&](int buf, u32x4(&fa)[3][TM], u32x4(&fb)[3][TN]) {
    const unsigned* as = As + buf * 3 * BM * SLD + (wm * (BM / 2) + li) * SLD + rcol;
    const unsigned* bs = Bs + buf * 3 * BN * SLD + (wn * (BN / 2) + li) * SLD + rcol;
#pragma unroll
    for (int pc = 0; pc < 3; ++pc) {
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[pc][i] = *(const u32x4*)(as + (pc * BM + i * 32) * SLD);
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[pc][j] = *(const u32x4*)(bs + (pc * BN + j * 32) * SLD);
    }
  };

  load_tile(ra0, rb0);       // tile 0
  load_tile(ra1, rb1);       // tile 1
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) asm volatile("" : "+a"(acc[i][j]));  // (64 accumulator writes: here, not in front of the loop)
  __builtin_amdgcn_sched_barrier(0);
  store_tile(0, ra0, rb0);
  __syncthreads();
  load_tile(ra0, rb0);       // tile 2
  read_frags(0, fa0, fb0);   // tile 0
  store_tile(1, ra1, rb1);
  __syncthreads();

  // iteration t: MFMAs on F_cur (tile t); reads tile t+1 from LDS[(t+1)&1] into F_nxt; splits R_cv (tile t+2) into
  // LDS[t&1]; loads tile t+3 into R_ld (t enters as its parity)
  auto k_step = [&](int t, const u32x4(&fa)[3][TM], const u32x4(&fb)[3][TN], u32x4(&na)[3][TM], u32x4(&nb)[3][TN],
                    const float4(&cv_a)[RA], const float4(&cv_b)[RB], float4(&ld_a)[RA], float4(&ld_b)[RB]) {
    const int bw_ = t & 1, br_ = bw_ ^ 1;
    const unsigned* as = As + br_ * 3 * BM * SLD + (wm * (BM / 2) + li) * SLD + rcol;
    const unsigned* bs = Bs + br_ * 3 * BN * SLD + (wn * (BN / 2) + li) * SLD + rcol;
    unsigned* aw = As + bw_ * 3 * BM * SLD + r0 * SLD + wcol;
    unsigned* bw = Bs + bw_ * 3 * BN * SLD + r0 * SLD + wcol;
    constexpr int PA[6] = {0, 0, 1, 1, 0, 2};
    constexpr int PB[6] = {0, 1, 0, 1, 2, 0};
    unsigned hb[NF][4], mb[NF][4], lb[NF][4];
    float r1[NF][4];
    uint2 hp[NF], mp[NF], lp[NF];
    unsigned ld_off[NF];
    next_tap();  // (scalar branch, before the fenced stream)
    __builtin_amdgcn_sched_barrier(0);
    static_for<0, NM>([&](auto qc) {
      constexpr int q = decltype(qc)::value;
      constexpr int pq = q / (TM * TN), ti = (q / TN) % TM, tj = q % TN;
      acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[PA[pq]][ti]),
                                                            __builtin_bit_cast(bf16x8, fb[PB[pq]][tj]), acc[ti][tj], 0, 0, 0);
      asm volatile("" : "+a"(acc[ti][tj]));  // (the MFMA opens its gap: the items below are issued in its shadow)
      if constexpr (q == SCH::SB) __syncthreads();  // (see StepSched::SB)
      static_for<0, SCH::NIT>([&](auto ic) {
        constexpr int it = decltype(ic)::value;
        constexpr int kind = kStepSched<NM, NF, NR, NCVF, NBW>.kind[it], ix = kStepSched<NM, NF, NR, NCVF, NBW>.idx[it];
        if constexpr (kStepSched<NM, NF, NR, NCVF, NBW>.gap[it] != q) {
        } else if constexpr (kind == 0) {
          // filters first: they are L2 hits and in front of the in-order load counter, the activation rows (possible
          // HBM misses) behind them -- and the split below takes the filters first, so an activation row has until the
          // middle of the NEXT step to land
          // load order: fp32 filters first (they are split first); with pre-split filters the activation rows first
          constexpr int f = ix / 2, half = ix % 2;
          constexpr bool isb = BPRE ? f >= RA : f < RB;
          constexpr int j = BPRE ? (isb ? f - RA : f) : (isb ? f : f - RB);
          if constexpr (half == 0) {
            unsigned off = isb ? (lt_k0 < b_lim[isb ? j : 0] ? b_cur[isb ? j : 0] : OOB)
                               : (lt_k0 < a_lim[isb ? 0 : j] ? a_cur[isb ? 0 : j] : OOB);
            asm volatile("" : "+v"(off));
            ld_off[f] = off;
          } else {
            if constexpr (isb) {
              ld_b[isb ? j : 0] = ldg_b128(rb_src, ld_off[f]);
              b_cur[isb ? j : 0] += B_STEP;
            } else {
              ld_a[isb ? 0 : j] = ldg_b128(ra_src, ld_off[f]);
              a_cur[isb ? 0 : j] += SBK * 4;
            }
            if constexpr (f == NF - 1) {
              lt_k0 += SBK;
              lt_cin += SBK;
            }
          }
        } else if constexpr (kind == 1) {
          constexpr int r = ix;  // planes in order h, m, l; within a plane A fragments then B fragments
          constexpr int pc = r / (TM + TN), x = r % (TM + TN);
          if constexpr (x < TM) na[pc][x < TM ? x : 0] = *(const u32x4*)(as + (pc * BM + x * 32) * SLD);
          else nb[pc][x < TM ? 0 : x - TM] = *(const u32x4*)(bs + (pc * BN + (x - TM) * 32) * SLD);
        } else if constexpr (kind == 3) {
          // pre-split filter chunk: as loaded
          *(float4*)(Bs + bw_ * 3 * BN * SLD + b_lds[ix]) = cv_b[ix];
        } else {
          constexpr int fo = ix / 13, r = ix % 13;
          // order of work: B floats, then A floats (f indexes A then B); pre-split filters: A floats only
          constexpr int f = BPRE ? fo : (fo < RB ? RA + fo : fo - RB);
          const float4 v = f < RA ? cv_a[f < RA ? f : 0] : cv_b[f < RA ? 0 : f - RA];
          constexpr int e0 = (r & 1) * 2, e1 = e0 + 1;  // the two elements of items 0..7
          if constexpr (r < 2) {  // H: top 16 bits
            const float x0 = e0 == 0 ? v.x : v.z, x1 = e0 == 0 ? v.y : v.w;
            hb[f][e0] = __float_as_uint(x0) & 0xffff0000u;
            hb[f][e1] = __float_as_uint(x1) & 0xffff0000u;
            asm volatile("" : "+v"(hb[f][e0]), "+v"(hb[f][e1]));
          } else if constexpr (r < 4) {  // R: x - h, exact
            const float x0 = e0 == 0 ? v.x : v.z, x1 = e0 == 0 ? v.y : v.w;
            r1[f][e0] = x0 - __uint_as_float(hb[f][e0]);
            r1[f][e1] = x1 - __uint_as_float(hb[f][e1]);
            asm volatile("" : "+v"(r1[f][e0]), "+v"(r1[f][e1]));
          } else if constexpr (r < 6) {  // M: top 16 bits of the remainder
            mb[f][e0] = __float_as_uint(r1[f][e0]) & 0xffff0000u;
            mb[f][e1] = __float_as_uint(r1[f][e1]) & 0xffff0000u;
            asm volatile("" : "+v"(mb[f][e0]), "+v"(mb[f][e1]));
          } else if constexpr (r < 8) {  // L: exact; <= 8 significant bits
            lb[f][e0] = __float_as_uint(r1[f][e0] - __uint_as_float(mb[f][e0]));
            lb[f][e1] = __float_as_uint(r1[f][e1] - __uint_as_float(mb[f][e1]));
            asm volatile("" : "+v"(lb[f][e0]), "+v"(lb[f][e1]));
          } else if constexpr (r == 8) {  // pack the upper halves of two dwords: bytes {S1.2, S1.3, S0.2, S0.3}
            hp[f].x = __builtin_amdgcn_perm(hb[f][1], hb[f][0], 0x07060302u);
            hp[f].y = __builtin_amdgcn_perm(hb[f][3], hb[f][2], 0x07060302u);
            asm volatile("" : "+v"(hp[f].x), "+v"(hp[f].y));
          } else if constexpr (r == 9) {
            mp[f].x = __builtin_amdgcn_perm(mb[f][1], mb[f][0], 0x07060302u);
            mp[f].y = __builtin_amdgcn_perm(mb[f][3], mb[f][2], 0x07060302u);
            asm volatile("" : "+v"(mp[f].x), "+v"(mp[f].y));
          } else if constexpr (r == 10) {
            lp[f].x = __builtin_amdgcn_perm(lb[f][1], lb[f][0], 0x07060302u);
            lp[f].y = __builtin_amdgcn_perm(lb[f][3], lb[f][2], 0x07060302u);
            asm volatile("" : "+v"(lp[f].x), "+v"(lp[f].y));
          } else {
            unsigned* w = f < RA ? aw + 64 * f * SLD : bw + 64 * (f - RA) * SLD;
            constexpr int rows = f < RA ? BM : BN;
            if constexpr (r == 11) {
              *(uint2*)(w + 0 * rows * SLD) = hp[f];
              *(uint2*)(w + 1 * rows * SLD) = mp[f];
            } else {
              *(uint2*)(w + 2 * rows * SLD) = lp[f];
            }
          }
        }
      });
      __builtin_amdgcn_sched_barrier(0);
    });
  };
  // always whole pairs of K-steps (an odd count runs one extra all-zero step): no conditional between the two halves,
  // so the register sets swap roles without copies
  const unsigned long long t_loop = p.trace ? __builtin_readcyclecounter() : 0ull;
  for (int t = 0; t < nk; t += 2) {
    k_step(0, fa0, fb0, fa1, fb1, ra0, rb0, ra1, rb1);
    k_step(1, fa1, fb1, fa0, fb0, ra1, rb1, ra0, rb0);
  }
  if constexpr (FUSE || ELDS) __syncthreads();  // (the last steps' staging writes / fragment reads vs the epilogue's use of the same LDS)
  const unsigned long long t_loop_end = p.trace ? __builtin_readcyclecounter() : 0ull;

  if constexpr (FUSE) {
    // ---- fused bottleneck tail (resnet.py:92-100): o2 = relu(bn2(conv2(x))) never leaves the CU. The tile's 128 x 64
    // values are split into the three bf16 planes exactly as the K loop of a separate 1x1 launch would split them, laid
    // out in LDS as that launch's four K-steps of A (rows of 32 B, halves swizzled: SLD), and contracted against the
    // pre-split 1x1 weights N2 / 64 column chunks at a time -- B fragments straight from L2 (a lane's 16 bytes are k
    // 8*lh.. of filter row n in the [K/16][3][N][16] layout: 1 KB contiguous per wave and plane), same K-step and product
    // order as igemm_split_kernel's loop, so the result is bit-identical to the two-launch form. The second conv's
    // epilogue (bn3, + residual, ReLU) works on the accumulator layout directly: 32 lanes = 128 contiguous bytes of a row.
    static_assert(BM == 128 && BN == 64 && !STEM && BPRE, "fused tail: 128 x 64 tile over pre-split weights");
    unsigned short* A2 = (unsigned short*)smem;  // [4 K-steps][3 planes][BM rows][16] bf16
    {
      const int nn = wn * 32 + li;  // this lane's column of the 64 (N == BN)
      const float sc1 = p.scale ? p.scale[nn] : 1.f, sh1 = p.shift ? p.shift[nn] : 0.f;
      const int ks = nn >> 4, kk = nn & 15;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          float v = acc[i][0][r] * sc1 + sh1;
          if (p.relu) v = fmaxf(v, 0.f);
          const unsigned hb = __float_as_uint(v) & 0xffff0000u;
          const float r1 = v - __uint_as_float(hb);
          const unsigned mb = __float_as_uint(r1) & 0xffff0000u;
          const unsigned lb = __float_as_uint(r1 - __uint_as_float(mb));
          unsigned short* w = A2 + ((ks * 3) * BM + row) * 16 + (((kk >> 3) ^ swz(row)) * 8 + (kk & 7));
          w[0 * BM * 16] = (unsigned short)(hb >> 16);
          w[1 * BM * 16] = (unsigned short)(mb >> 16);
          w[2 * BM * 16] = (unsigned short)(lb >> 16);
        }
    }
    __syncthreads();
    const int N2 = p.N2, nch = N2 / 64;
    const unsigned* a_rd = (const unsigned*)smem + (wm * (BM / 2) + li) * SLD + rcol;
    // every global access of this phase is a buffer access = descriptor (uniform) + 32-bit lane offset (the lane's own
    // part plus a uniform term: one v_add with an SGPR operand): no 64-bit address per accumulator element. Rows past M
    // fall outside the descriptors' ranges -- their loads return zeros, their stores are dropped. (The whole offset is
    // in the VGPR operand: the instruction's SGPR offset is NOT part of the hardware's range check.)
    const int rows_valid = (p.M - m0) < BM ? (p.M - m0) : BM;
    const __amdgpu_buffer_rsrc_t rc_dst =
        __builtin_amdgcn_make_buffer_rsrc((void*)(Cb + (long)m0 * p.ldc), 0, (int)(rows_valid * p.ldc * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rc_res = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.residual ? p.residual + (long)m0 * p.ldr : p.A), 0, p.residual ? (int)(rows_valid * p.ldr * 4) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rc_b2 = __builtin_amdgcn_make_buffer_rsrc((void*)p.Bw2, 0, 12 * N2 * 32, 0x00020000);
    const int ldc4 = (int)p.ldc * 4, ldr4 = (int)p.ldr * 4;
    const int lrow = wm * (BM / 2) + 4 * lh, lcol4 = (wn * 32 + li) * 4;
    const int vo_c = lrow * ldc4 + lcol4, vo_r = lrow * ldr4 + lcol4;
    const int vo_b = (wn * 32 + li) * 32 + lh * 16;
    u32x4 bfr[4][3];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) bfr[ks][pl] = __builtin_amdgcn_raw_buffer_load_b128(rc_b2, vo_b + (ks * 3 + pl) * N2 * 32, 0, 0);
    constexpr int PA[6] = {0, 0, 1, 1, 0, 2};
    constexpr int PB[6] = {0, 1, 0, 1, 2, 0};
    for (int c = 0; c < nch; ++c) {
      const int nn = c * 64 + wn * 32 + li;
      const float sc2 = p.scale2 ? p.scale2[nn] : 1.f, sh2 = p.shift2 ? p.shift2[nn] : 0.f;
      float res[TM][16];
      int vr = vo_r + c * 256, vc = vo_c + c * 256;  // (opaque per chunk: 64 loop-invariant row addresses would otherwise
      asm volatile("" : "+v"(vr), "+v"(vc));         //  be kept in registers across the loop -- one workgroup less per CU)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          res[i][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                    rc_res, vr + (i * 32 + (r & 3) + 8 * (r >> 2)) * ldr4, 0, 0));
      f32x16 acc2[TM];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[i][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        u32x4 afr[3][TM];
#pragma unroll
        for (int pc = 0; pc < 3; ++pc)
#pragma unroll
          for (int i = 0; i < TM; ++i) afr[pc][i] = *(const u32x4*)(a_rd + ((ks * 3 + pc) * BM + i * 32) * SLD);
#pragma unroll
        for (int pq = 0; pq < 6; ++pq)
#pragma unroll
          for (int i = 0; i < TM; ++i)
            acc2[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, afr[PA[pq]][i]),
                                                              __builtin_bit_cast(bf16x8, bfr[ks][PB[pq]]), acc2[i], 0, 0, 0);
        // the next chunk's filter fragments of this K-step: their registers are free now (past the last chunk the
        // offsets fall outside the descriptor: zeros, unused)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
          bfr[ks][pl] = __builtin_amdgcn_raw_buffer_load_b128(rc_b2, vo_b + ((ks * 3 + pl) * N2 + (c + 1) * 64) * 32, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = acc2[i][r] * sc2 + sh2;
          v += res[i][r];
          if (p.relu2) v = fmaxf(v, 0.f);
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rc_dst,
                                                vc + (i * 32 + (r & 3) + 8 * (r >> 2)) * ldc4, 0, 0);
        }
    }
    return;
  }

  // ---- epilogue on the accumulator registers (default): epilogue_regs above
  if constexpr (!ELDS) {
    epilogue_regs<BM, BN>(p, acc, m0, n0, Cb, wm, wn, li, lh);
    if (p.trace && tid == 0) {
      unsigned long long* tr = p.trace + ((long)blockIdx.z * gridDim.x + blockIdx.x) * 8;
      tr[0] = t_start;
      tr[1] = t_loop;
      tr[2] = t_loop_end;
      tr[3] = __builtin_readcyclecounter();
      tr[4] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);  // HW_ID | XCC_ID
      tr[5] = wall_clock64();
      tr[6] = w_start;
      tr[7] = 0;
    }
    return;
  }

  // ---- epilogue through LDS (same C/D map as the f32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)) ----
  // The residual (and ReLU-adjoint mask) rows of the whole tile are requested BEFORE the accumulators go through LDS:
  // the 1x1 expand convs (K = 64..512) are HBM-bound on exactly these reads and the output writes, and a load issued
  // per pass would put one HBM round trip in front of every store.
  constexpr int TPR = BN / 4;
  constexpr int RPP = 256 / TPR;
  constexpr int NP = BM / RPP;  // passes
  const int ec = (tid % TPR) * 4;
  const int er = tid / TPR;
  const int n = n0 + ec;
  const bool full4 = p.vec_io && (n + 3) < p.N;
  // A tile lies in ONE geometry segment except for the single tile row that straddles M0: there the output / residual
  // rows are a base pointer plus a constant step per pass (the general form costs two 64-bit multiply-adds and a select
  // per pass and operand -- a few thousand cycles of a short-K tile's life with one wave per SIMD).
  const bool one_seg = m0 + BM <= p.M0 || m0 >= p.M0;  // (uniform)
  const bool seg1 = m0 >= p.M0;
  const long ld_c = seg1 ? p.ldc1 : p.ldc, ld_r = seg1 ? p.ldr1 : p.ldr;
  const int mrel = (seg1 ? m0 - p.M0 : m0) + er;
  float* const c_base = (seg1 ? p.C1 : Cb) + (long)mrel * ld_c + n;
  const float* const r_base = (seg1 ? p.residual1 : p.residual) + (long)mrel * ld_r + n;
  float4 rres[NP];
  if (p.residual && full4) {
    if (one_seg) {
#pragma unroll
      for (int q = 0; q < NP; ++q)
        rres[q] = (m0 + er + q * RPP < p.M) ? *(const float4*)(r_base + (long)(q * RPP) * ld_r) : make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
#pragma unroll
      for (int q = 0; q < NP; ++q) {
        const int m = m0 + er + q * RPP;
        rres[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (m < p.M) {
          const bool s1 = m >= p.M0;
          rres[q] = *(const float4*)(s1 ? p.residual1 + (long)(m - p.M0) * p.ldr1 + n : p.residual + (long)m * p.ldr + n);
        }
      }
    }
  }
  // The common tile -- all BM rows and BN columns inside the problem, one geometry segment, vector I/O
  // -- takes a straight-line epilogue (uniform choice): two float4 loads for scale / shift, every pass's LDS read
  // issued before the first is used, no per-pass bounds / flag branches. (The general form below loads the eight scale /
  // shift values one by one, each behind its own wait, and serialises the passes behind their LDS reads.)
  const bool fast_epi = p.vec_io && p.vec_ss && one_seg && m0 + BM <= p.M && n0 + BN <= p.N;
  // The ReLU-adjoint mask rows of a data-gradient launch (activation rows: only their signs matter) are requested ahead
  // like the residual rows -- a load per pass inside the store loop cannot be hoisted over the stores (possible alias)
  // and costs an HBM round trip per pass --, but in halves of at most 8 passes: the first half here, the second when the
  // first has been turned into sign bits, so that a 128 x 128 tile stays inside its two-blocks-per-CU register budget.
  constexpr int EH = NP > 8 ? 2 : 1, NH = NP / EH;
  float4 rmask[NH];
  const float* const k_base = p.mask ? p.mask + (long)(m0 + er) * p.ldm + n : nullptr;
  if (p.mask && fast_epi) {
#pragma unroll
    for (int q = 0; q < NH; ++q) rmask[q] = *(const float4*)(k_base + (long)(q * RPP) * p.ldm);
  }
  float sc[4], sh[4];  // (requested with the residual rows: in flight while the accumulators go through LDS)
  if (fast_epi) {
    const float4 s4 = p.scale ? *(const float4*)(p.scale + n) : make_float4(1.f, 1.f, 1.f, 1.f);
    const float4 h4 = p.shift ? *(const float4*)(p.shift + n) : make_float4(0.f, 0.f, 0.f, 0.f);
    sc[0] = s4.x; sc[1] = s4.y; sc[2] = s4.z; sc[3] = s4.w;
    sh[0] = h4.x; sh[1] = h4.y; sh[2] = h4.z; sh[3] = h4.w;
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const bool nok = (n + q) < p.N;
      sc[q] = (nok && p.scale) ? p.scale[n + q] : 1.f;
      sh[q] = (nok && p.shift) ? p.shift[n + q] : 0.f;
    }
  }
  float* Cs = smem;  // [BM][CLD]
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      float* cw = Cs + (wm * (BM / 2) + i * 32 + 4 * lh) * CLD + wn * (BN / 2) + j * 32 + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) cw[((r & 3) + 8 * (r >> 2)) * CLD] = acc[i][j][r];
    }
  __syncthreads();  // (waits for the LDS writes only: the residual / scale / shift loads stay in flight)
#pragma unroll
  for (int q = 0; q < 4; ++q) sc[q] *= p.alpha;
  if (fast_epi) {
    auto emit = [&](auto res_c, auto relu_c, auto mask_c) {
      constexpr bool RES = decltype(res_c)::value != 0, RELU = decltype(relu_c)::value != 0, MASK = decltype(mask_c)::value != 0;
#pragma unroll
      for (int h = 0; h < EH; ++h) {
        float4 a4[NH];  // (every pass's LDS read of the half issued before the first is used)
#pragma unroll
        for (int q = 0; q < NH; ++q) a4[q] = *(const float4*)(Cs + (er + (h * NH + q) * RPP) * CLD + ec);
        unsigned mb = 0;
        if (MASK) {
#pragma unroll
          for (int q = 0; q < NH; ++q)
            mb |= ((rmask[q].x > 0.f ? 1u : 0u) | (rmask[q].y > 0.f ? 2u : 0u) | (rmask[q].z > 0.f ? 4u : 0u) |
                   (rmask[q].w > 0.f ? 8u : 0u)) << (4 * q);
          if (h + 1 < EH) {  // (the next half's rows: they land while this half is stored)
#pragma unroll
            for (int q = 0; q < NH; ++q) rmask[q] = *(const float4*)(k_base + (long)(((h + 1) * NH + q) * RPP) * p.ldm);
          }
        }
#pragma unroll
        for (int q = 0; q < NH; ++q) {
          const int qq = h * NH + q;
          float v[4] = {a4[q].x * sc[0] + sh[0], a4[q].y * sc[1] + sh[1], a4[q].z * sc[2] + sh[2], a4[q].w * sc[3] + sh[3]};
          if (RES) {
            v[0] += rres[qq].x;
            v[1] += rres[qq].y;
            v[2] += rres[qq].z;
            v[3] += rres[qq].w;
          }
          if (RELU) {
#pragma unroll
            for (int t = 0; t < 4; ++t) v[t] = fmaxf(v[t], 0.f);
          }
          if (MASK) {
            const unsigned nb = mb >> (4 * q);
            v[0] = (nb & 1u) ? v[0] : 0.f;
            v[1] = (nb & 2u) ? v[1] : 0.f;
            v[2] = (nb & 4u) ? v[2] : 0.f;
            v[3] = (nb & 8u) ? v[3] : 0.f;
          }
          *(float4*)(c_base + (long)(qq * RPP) * ld_c) = make_float4(v[0], v[1], v[2], v[3]);
        }
      }
    };
    if (p.mask) {  // (data gradients)
      if (p.residual) {
        if (p.relu) emit(IC<1>{}, IC<1>{}, IC<1>{});
        else emit(IC<1>{}, IC<0>{}, IC<1>{});
      } else {
        if (p.relu) emit(IC<0>{}, IC<1>{}, IC<1>{});
        else emit(IC<0>{}, IC<0>{}, IC<1>{});
      }
    } else if (p.residual) {
      if (p.relu) emit(IC<1>{}, IC<1>{}, IC<0>{});
      else emit(IC<1>{}, IC<0>{}, IC<0>{});
    } else {
      if (p.relu) emit(IC<0>{}, IC<1>{}, IC<0>{});
      else emit(IC<0>{}, IC<0>{}, IC<0>{});
    }
  } else if (full4) {
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      const int rr = er + q * RPP;
      const int m = m0 + rr;
      if (m < p.M) {
        const float4 a4 = *(const float4*)(Cs + rr * CLD + ec);
        float v[4] = {a4.x * sc[0] + sh[0], a4.y * sc[1] + sh[1], a4.z * sc[2] + sh[2], a4.w * sc[3] + sh[3]};
        float* cp = c_base + (long)(q * RPP) * ld_c;
        if (!one_seg) {
          const bool s1 = m >= p.M0;
          cp = s1 ? p.C1 + (long)(m - p.M0) * p.ldc1 + n : Cb + (long)m * p.ldc + n;
        }
        if (p.residual) {
          v[0] += rres[q].x;
          v[1] += rres[q].y;
          v[2] += rres[q].z;
          v[3] += rres[q].w;
        }
        if (p.relu) {
#pragma unroll
          for (int t = 0; t < 4; ++t) v[t] = fmaxf(v[t], 0.f);
        }
        if (p.mask) {
          const float4 k4 = *(const float4*)(p.mask + (long)m * p.ldm + n);
          v[0] = k4.x > 0.f ? v[0] : 0.f;
          v[1] = k4.y > 0.f ? v[1] : 0.f;
          v[2] = k4.z > 0.f ? v[2] : 0.f;
          v[3] = k4.w > 0.f ? v[3] : 0.f;
        }
        *(float4*)cp = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
  } else {
    for (int q = 0; q < NP; ++q) {
      const int rr = er + q * RPP;
      const int m = m0 + rr;
      if (m >= p.M) break;
      const bool s1 = m >= p.M0;
      float* cp = s1 ? p.C1 + (long)(m - p.M0) * p.ldc1 + n : Cb + (long)m * p.ldc + n;
      const float* rp = s1 ? p.residual1 + (long)(m - p.M0) * p.ldr1 + n : p.residual + (long)m * p.ldr + n;
      for (int t = 0; t < 4; ++t)
        if ((n + t) < p.N) {
          float x = Cs[rr * CLD + ec + t] * sc[t] + sh[t];
          if (p.residual) x += rp[t];
          if (p.relu) x = fmaxf(x, 0.f);
          if (p.mask && !(p.mask[(long)m * p.ldm + n + t] > 0.f)) x = 0.f;
          cp[t] = x;
        }
    }
  }
  if (p.trace && tid == 0) {
    unsigned long long* tr = p.trace + ((long)blockIdx.z * gridDim.x + blockIdx.x) * 8;
    tr[0] = t_start;
    tr[1] = t_loop;
    tr[2] = t_loop_end;
    tr[3] = __builtin_readcyclecounter();
    tr[4] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);  // HW_ID | XCC_ID
    tr[5] = wall_clock64();
    tr[6] = w_start;
    tr[7] = 0;
  }
}

// The kernels proper. The 128 x 128 tile is compiled for TWO workgroups per CU: 192 arch VGPRs beside its 64 accumulator
// registers (the compiler, left alone with the one-block bound, spends 200+ on the epilogue's prefetches and halves the
// occupancy of the whole K loop).
template <int BM, int BN, int STEM, int BPRE = 0, int FUSE = 0, int ELDS = 0>
__global__ void __launch_bounds__(256, 2) igemm_split_kernel(IgemmParams p) {
  igemm_split_body<BM, BN, STEM, BPRE, FUSE, ELDS>(p);
}
template <int STEM, int BPRE, int ELDS>
__global__ void __launch_bounds__(256, 1) __attribute__((amdgpu_num_vgpr(192))) igemm_split_kernel_128(IgemmParams p) {
  igemm_split_body<128, 128, STEM, BPRE, 0, ELDS>(p);
}

// ---- the three-workgroups-per-CU contraction kernel (igemm_dma_kernel): weights by LDS-DMA, single fragment set --------------
// Round 5. igemm_split_kernel_128 keeps two fragment sets and two staging register sets per operand in flight (224 registers:
// two workgroups per CU) and its tiles spend as long in prologue + epilogue as in a short K loop. This kernel gives the
// registers back to occupancy:
//  * the pre-split WEIGHT planes go from L2 straight into the LDS planes by LDS-DMA (buffer_load_dwordx4 ... lds: a wave
//    instruction moves 32 rows x 32 bytes = 1 KB of one plane; the LDS image is lane-linear, so the 16-byte halves of a row
//    are swizzled on the SOURCE address): no staging registers, no staging writes for B;
//  * ONE fragment set: a step reads its 12 fragments behind the barrier that publishes the tile, then issues its 24 MFMAs
//    -- with three waves per SIMD the other workgroups' MFMAs run in that shadow;
//  * the activation rows (APRE = 0: fp32 rows, any conv geometry of igemm_split_body) keep ONE staging register set: tile
//    t + 2's rows, requested a step ago, are split and written behind step t's second barrier, then tile t + 3's are requested;
//  * APRE = 1: the activation rows arrive as the three bf16 planes of their exact split ([Kp/16][3][rows][16], the weight
//    layout: written by a producer's epilogue or by dana_split_weight) and take the DMA path as well: no fp32 operand left
//    to split, no VALU between the matrix instructions, and the pipeline depth is a template parameter (NST stages).
//  * epilogue on the accumulator registers (epilogue_regs), residual rows requested in halves.
//  NST = 2: 49.2 KB of LDS, <= 168 registers: THREE workgroups per CU; two barriers per K-step (tile published / stage free).
//  NST >= 3 (APRE only): one barrier per K-step, NST - 1 tiles in flight (3: two workgroups per CU).
//  DF (APRE, NST >= 3): TWO fragment sets -- a tile-starved launch has one workgroup per CU and nothing to run in the shadow
//  of a step's fragment reads (a lone workgroup walks K at ~1 200 cycles per step either way: 768 of MFMA + the read round
//  trip); with the next tile's fragments read between this tile's MFMAs the step is the MFMA time. ~190 registers.
// Same K-step order, same split, the same six products in the same order as igemm_split_kernel -> the same bits.
template <int BN, int NST, int APRE, int DF = 0>
__global__ void __launch_bounds__(256, NST == 2 ? 3 : (NST == 3 ? 2 : 1)) igemm_dma_kernel(IgemmParams p) {
  static_assert(APRE || NST == 2, "fp32 activation rows: two stages");
  static_assert(!DF || (APRE && NST >= 3), "two fragment sets: planes x planes, three or more stages");
  constexpr int BM = 128, TM = 2, TN = BN / 64;
  constexpr int STAGE_B = 3 * (BM + BN) * 32;            // bytes per stage
  constexpr int NPB = 3 * BN / 32, PB = (NPB + 3) / 4;   // B pieces per step, per wave
  constexpr int NDMA = (APRE ? 3 : 0) + PB;              // DMA instructions per wave and K-step
  constexpr int AHEAD = NST == 2 ? 2 : NST - 1;          // tiles requested ahead of the one being multiplied
  constexpr int RA = BM / 64;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  const unsigned long long t_start = p.trace ? __builtin_readcyclecounter() : 0ull;
  const unsigned long long w_start = p.trace ? wall_clock64() : 0ull;
  const int vid = xcd_remap((int)(blockIdx.z * gridDim.x + blockIdx.x), (int)(gridDim.x * gridDim.z));
  const int zb = vid / (int)gridDim.x;
  const int tile = vid - zb * (int)gridDim.x;
  const int tm_idx = tile / p.tiles_n, tn_idx = tile % p.tiles_n;
  const int m0 = tm_idx * BM, n0 = tn_idx * BN;
  float* Cb = p.C + (long)zb * p.batch_c;
  __amdgpu_buffer_rsrc_t ra_src =
      __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + (long)zb * p.batch_a), 0, (int)p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb_src =
      __builtin_amdgcn_make_buffer_rsrc((void*)(p.Bw + (long)zb * p.batch_b), 0, (int)p.b_bytes, 0x00020000);
  const int nk = (p.K + SBK - 1) / SBK;

  // ---- DMA pieces of this wave: B pieces q = wave + 4 j -> (plane, row group) (64-wide tiles: six pieces on eight slots -- a
  // slot past the end repeats the wave's first piece: same bytes, same place); APRE: A rows 32 * wave .. + 31 of the three
  // planes. Lane i of a piece holds the 16-byte slot i & 1 of row i >> 1, i.e. the half (i & 1) ^ swz(row) of its 32 bytes.
  const int prow = lane >> 1, phalf = (lane & 1) ^ ((lane >> 4) & 1);
  const int arow = m0 + 32 * wave + prow;
  const unsigned a_vo = arow < p.M ? (unsigned)(arow * 32 + phalf * 16) : OOB;
  const unsigned a_pstride = (unsigned)p.a_rows * 32u, b_pstride = (unsigned)p.N * 32u;  // bytes between planes / K-step thirds
  unsigned b_vo[PB];
  int b_pl[PB], b_lds[PB];
#pragma unroll
  for (int j = 0; j < PB; ++j) {
    int q = wave + 4 * j;
    if (q >= NPB) q -= 4;
    const int pl = q / (BN / 32), rg = q % (BN / 32);
    const int n = n0 + 32 * rg + prow;
    b_vo[j] = n < p.N ? (unsigned)(n * 32 + phalf * 16) : OOB;
    b_pl[j] = pl;
    b_lds[j] = 3 * BM * 32 + (pl * BN + 32 * rg) * 32;
  }
  typedef __attribute__((address_space(3))) void lds_void;
  typedef __attribute__((address_space(3))) char lds_char;
  lds_char* const lds0 = (lds_char*)(lds_void*)smem;
  auto issue = [&](int t, int stage) {  // tile t -> stage (all pieces out of range past the last K-step: zeros, never read)
    const bool live = t < nk;
    lds_char* base = lds0 + stage * STAGE_B;
    if constexpr (APRE) {
      const unsigned av = live ? a_vo : OOB;
      const unsigned sa = (unsigned)t * 3u * a_pstride;
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra_src, (lds_void*)(base + (pl * BM + 32 * wave) * 32), 16, (int)av,
                                                 (int)(sa + pl * a_pstride), 0, 0);
    }
    const unsigned sb = (unsigned)t * 3u * b_pstride;
#pragma unroll
    for (int j = 0; j < PB; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb_src, (lds_void*)(base + b_lds[j]), 16, (int)(live ? b_vo[j] : OOB),
                                               (int)(sb + b_pl[j] * b_pstride), 0, 0);
  };

  // ---- fp32 activation rows (APRE = 0): igemm_split_body's K walk -- "taps" of cin_t channels, one 32-bit byte offset and a
  // tap-validity mask per staged row, a per-lane K limit that is DEAD for padding taps / missing rows / the K tail (the load
  // then takes an out-of-range offset: zeros), optional second geometry segment and second K segment (A2)
  int cin_t = p.Cin;
  const int kw_t = p.KW, kh_t = p.KH;
  const int lda4 = p.lda * 4;
  const int c4 = tid & 3, r0 = tid >> 2;
  const int wcol = ((c4 >> 1) ^ swz(r0)) * 4 + (c4 & 1) * 2;
  const int klim = p.K - c4 * 4;
  constexpr int DEAD = (int)0x80000000;
  unsigned a_off[RA], a_mask[RA], a_off2[RA];
  int a_dseg[RA];
  int lt_k0 = 0, lt_cin = 0, lt_kh = 0, lt_kw = 0;
  unsigned lt_bit = 1u;
  unsigned a_cur[RA];
  int a_lim[RA];
  if constexpr (!APRE) {
    const bool lin = p.KH * p.KW == 1 && p.stride == 1 && p.pad == 0;
#pragma unroll
    for (int j = 0; j < RA; ++j) {
      const int m = m0 + r0 + 64 * j;
      const bool ok = m < p.M;
      const bool s1 = ok && m >= p.M0;
      const int mm = ok ? (s1 ? m - p.M0 : m) : 0;
      if (lin && !p.A2) {
        a_off[j] = (unsigned)(((s1 ? p.pix1 : 0) + mm) * lda4 + c4 * 16);
        a_mask[j] = ok ? 1u : 0u;
        a_dseg[j] = 0;
        a_off2[j] = OOB;
      } else {
        const int IH = s1 ? p.IH1 : p.IH, IW = s1 ? p.IW1 : p.IW, OW = s1 ? p.OW1 : p.OW;
        const int ohw = (s1 ? p.OH1 : p.OH) * OW;
        const int img = mm / ohw, rem = mm - img * ohw;
        const int oh = rem / OW, ow = rem - oh * OW;
        const int ih0 = oh * p.stride - p.pad, iw0 = ow * p.stride - p.pad;
        const int pix = (s1 ? p.pix1 : 0) + (img * IH + ih0) * IW;
        unsigned mask = 0;
        if (ok) {
          unsigned colbits = 0;
          for (int kw = 0; kw < kw_t; ++kw)
            if (iw0 + kw >= 0 && iw0 + kw < IW) colbits |= 1u << kw;
          for (int kh = 0; kh < kh_t; ++kh)
            if (ih0 + kh >= 0 && ih0 + kh < IH) mask |= colbits << (kh * kw_t);
        }
        a_off[j] = (unsigned)((pix + iw0) * lda4 + c4 * 16);
        a_mask[j] = mask;
        a_dseg[j] = s1 ? (p.IW1 - p.IW) * lda4 : 0;
        a_off2[j] = (p.A2 && ok) ? (unsigned)(((s1 ? p.pix21 : 0) + ((long)img * (s1 ? p.IH21 : p.IH2) + oh * p.stride2) * (s1 ? p.IW21 : p.IW2) +
                                              ow * p.stride2) * p.lda2 * 4 + c4 * 16)
                                 : OOB;
      }
      a_cur[j] = a_off[j];
      a_lim[j] = (a_mask[j] & 1u) ? klim : DEAD;
    }
  }
  auto next_tap = [&]() {  // rare: the tap's channels are used up
    if (lt_cin >= cin_t) {
      lt_cin = 0;
      if (p.A2) {
        // the second K segment: another tensor (own descriptor, own pixel map), K1 channels; nothing after it
        ra_src = __builtin_amdgcn_make_buffer_rsrc((void*)p.A2, 0, (int)p.a2_bytes, 0x00020000);
        cin_t = 0x7fffffff;
#pragma unroll
        for (int j = 0; j < RA; ++j) {
          a_cur[j] = a_off2[j];
          a_lim[j] = a_off2[j] != OOB ? klim : DEAD;
        }
        return;
      }
      lt_bit <<= 1;
      if (++lt_kw == kw_t) {
        lt_kw = 0;
        ++lt_kh;
      }
      const int d0 = lt_kh * p.IW * lda4 + lt_kw * lda4;
#pragma unroll
      for (int j = 0; j < RA; ++j) {
        a_cur[j] = a_off[j] + (unsigned)(d0 + lt_kh * a_dseg[j]);
        a_lim[j] = (a_mask[j] & lt_bit) ? klim : DEAD;
      }
    }
  };
  float4 ra[RA];
  auto load_a = [&]() {  // the next K-step's rows of this lane -> ra
    next_tap();
#pragma unroll
    for (int j = 0; j < RA; ++j) {
      unsigned off = lt_k0 < a_lim[j] ? a_cur[j] : OOB;
      asm volatile("" : "+v"(off));
      ra[j] = ldg_b128(ra_src, off);
      a_cur[j] += SBK * 4;
    }
    lt_k0 += SBK;
    lt_cin += SBK;
  };
  auto split_a = [&](int stage) {  // ra -> the three bf16 planes of A in `stage`
    unsigned* as = (unsigned*)smem + stage * (STAGE_B / 4) + r0 * SLD + wcol;
#pragma unroll
    for (int j = 0; j < RA; ++j) {
      uint2 h, m, l;
      split3(ra[j], h, m, l);
      *(uint2*)(as + (0 * BM + 64 * j) * SLD) = h;
      *(uint2*)(as + (1 * BM + 64 * j) * SLD) = m;
      *(uint2*)(as + (2 * BM + 64 * j) * SLD) = l;
    }
  };

  f32x16 acc[TM][TN];
  if constexpr (APRE) {
#pragma unroll
    for (int a = 0; a < AHEAD; ++a) issue(a, a % NST);
  } else {
    load_a();  // tile 0
    issue(0, 0);
    issue(1, 1);
  }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  if constexpr (!APRE) {
    split_a(0);  // (the compiler waits for tile 0's rows here)
    load_a();    // tile 1
    split_a(1);
    load_a();    // tile 2: split in step 0
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RA) : "memory");  // both tiles' B pieces landed (older than tile 2's row loads)
  } else {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((AHEAD - 1) * NDMA) : "memory");  // tile 0 landed (this wave's pieces)
  }
  // fragment addresses (dwords): row li of the wave's rows, 16-byte half lh (swizzled like the staging image)
  const int rcol = (lh ^ swz(li)) * 4;
  const unsigned* const a_rd = (const unsigned*)smem + (wm * (BM / 2) + li) * SLD + rcol;
  const unsigned* const b_rd = (const unsigned*)smem + 3 * BM * SLD + (wn * (BN / 2) + li) * SLD + rcol;
  constexpr int PA[6] = {0, 0, 1, 1, 0, 2};
  constexpr int PBp[6] = {0, 1, 0, 1, 2, 0};
  const unsigned long long t_loop = p.trace ? __builtin_readcyclecounter() : 0ull;
  if constexpr (DF) {
    // two fragment sets: step t multiplies tile t (set t & 1) while it reads tile t + 1 (published by the step's barrier) into
    // the other set; the barrier also retires every wave's reads of tile t (issued during step t - 1), whose stage takes tile
    // t + NST. In flight behind the barrier: tiles t + 2 .. t + NST.
    constexpr int U = (NST % 2) ? 2 * NST : NST;  // steps per loop trip: stage and fragment-set indices are static
    u32x4 fa0[3][TM], fb0[3][TN], fa1[3][TM], fb1[3][TN];
    issue(NST - 1, NST - 1);  // (the prologue issued tiles 0 .. NST - 2)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 1) * NDMA) : "memory");  // tile 0 landed (this wave's pieces)
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int pc = 0; pc < 3; ++pc) {
#pragma unroll
      for (int i = 0; i < TM; ++i) fa0[pc][i] = *(const u32x4*)(a_rd + (pc * BM + i * 32) * SLD);
#pragma unroll
      for (int j = 0; j < TN; ++j) fb0[pc][j] = *(const u32x4*)(b_rd + (pc * BN + j * 32) * SLD);
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * NDMA) : "memory");  // tile 1 landed
    auto step = [&](auto uc, const u32x4(&fa)[3][TM], const u32x4(&fb)[3][TN], u32x4(&na)[3][TM], u32x4(&nb)[3][TN], int t) {
      constexpr int u = decltype(uc)::value;
      constexpr int st_next = (u + 1) % NST, st_free = u % NST;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (this wave's reads of tile t: its stage is refilled below)
      __builtin_amdgcn_s_barrier();
      issue(t + NST, st_free);
      __builtin_amdgcn_sched_barrier(0);
      constexpr int NM = 6 * TM * TN, NR = 3 * (TM + TN);
      static_for<0, NM>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        constexpr int pq = q / (TM * TN), ti = (q / TN) % TM, tj = q % TN;
        acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[PA[pq]][ti]),
                                                              __builtin_bit_cast(bf16x8, fb[PBp[pq]][tj]), acc[ti][tj], 0, 0, 0);
        // one fragment read of tile t + 1 behind every second MFMA (planes in order h, m, l; A then B)
        if constexpr (q % 2 == 0 && q / 2 < NR) {
          constexpr int r = q / 2, pc = r / (TM + TN), x = r % (TM + TN);
          if constexpr (x < TM) na[pc][x < TM ? x : 0] = *(const u32x4*)(a_rd + st_next * (STAGE_B / 4) + (pc * BM + x * 32) * SLD);
          else nb[pc][x < TM ? 0 : x - TM] = *(const u32x4*)(b_rd + st_next * (STAGE_B / 4) + (pc * BN + (x - TM) * 32) * SLD);
        }
        __builtin_amdgcn_sched_barrier(0);
      });
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * NDMA) : "memory");  // tile t + 2 landed
    };
    for (int t0 = 0; t0 < nk; t0 += U) {
      static_for<0, U>([&](auto uc) {
        constexpr int u = decltype(uc)::value;
        if (t0 + u < nk) {  // (uniform)
          if constexpr (u % 2 == 0) step(uc, fa0, fb0, fa1, fb1, t0 + u);
          else step(uc, fa1, fb1, fa0, fb0, t0 + u);
        }
      });
    }
  } else
  for (int t0 = 0; t0 < nk; t0 += NST) {
    static_for<0, NST>([&](auto sc_) {
      constexpr int st = decltype(sc_)::value;
      const int t = t0 + st;
      if (t < nk) {  // (uniform)
        if constexpr (!APRE) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (this wave's staging writes of tile t)
        __builtin_amdgcn_s_barrier();  // tile t is in LDS for every wave; every wave is past its reads of tile t - 1
        if constexpr (NST >= 3) issue(t + AHEAD, (st + AHEAD) % NST);
        __builtin_amdgcn_sched_barrier(0);
        u32x4 fa[3][TM], fb[3][TN];
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) {
#pragma unroll
          for (int i = 0; i < TM; ++i) fa[pc][i] = *(const u32x4*)(a_rd + st * (STAGE_B / 4) + (pc * BM + i * 32) * SLD);
#pragma unroll
          for (int j = 0; j < TN; ++j) fb[pc][j] = *(const u32x4*)(b_rd + st * (STAGE_B / 4) + (pc * BN + j * 32) * SLD);
        }
        if constexpr (NST == 2) {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
          __builtin_amdgcn_s_barrier();  // every wave holds its fragments of tile t: the stage is free for tile t + 2
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (!APRE) {
            // tile t + 2's rows were requested a step ago, behind the B pieces of tile t + 1: the wait for them (vmcnt 0:
            // nothing younger is in flight) also retires those pieces. Then this step's requests: B pieces, next rows.
            split_a(st);
            __builtin_amdgcn_sched_barrier(0);
            issue(t + 2, st);
            load_a();  // tile t + 3
          } else {
            issue(t + 2, st);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int pq = 0; pq < 6; ++pq)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[PA[pq]][i]),
                                                                  __builtin_bit_cast(bf16x8, fb[PBp[pq]][j]), acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (APRE) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((AHEAD - 1) * NDMA) : "memory");  // tile t + 1 landed
      }
    });
  }
  const unsigned long long t_loop_end = p.trace ? __builtin_readcyclecounter() : 0ull;
  epilogue_regs<BM, BN, (NST == 2 && BN == 128) ? 1 : 0, 0>(p, acc, m0, n0, Cb, wm, wn, li, lh);
  if (p.trace && tid == 0) {
    unsigned long long* tr = p.trace + ((long)blockIdx.z * gridDim.x + blockIdx.x) * 8;
    tr[0] = t_start;
    tr[1] = t_loop;
    tr[2] = t_loop_end;
    tr[3] = __builtin_readcyclecounter();
    tr[4] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
    tr[5] = wall_clock64();
    tr[6] = w_start;
    tr[7] = 0;
  }
}

// ---- skinny GEMM (N <= 8): one wave per output row, lanes split K (RCNN_bbox_pred 2048->4,
// output_score_layer.linear2 1024->2: dana.py:246,304). HBM-bound on reading A once. ------------
template <int NMAX>
__global__ void __launch_bounds__(256)
gemm_skinny_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ c,
                   const float* __restrict__ scale, const float* __restrict__ shift, int M, int N, int K, long lda,
                   long ldb, long ldc, float alpha, int relu) {
  const int lane = threadIdx.x & 63;
  const long m = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  float acc[NMAX];
#pragma unroll
  for (int n = 0; n < NMAX; ++n) acc[n] = 0.f;
  const float4* ar = (const float4*)(a + m * lda);
  for (int k4 = lane; k4 < K / 4; k4 += 64) {
    const float4 x = ar[k4];
#pragma unroll
    for (int n = 0; n < NMAX; ++n)
      if (n < N) {
        const float4 w = ((const float4*)(b + n * ldb))[k4];
        acc[n] += x.x * w.x + x.y * w.y + x.z * w.z + x.w * w.w;
      }
  }
#pragma unroll
  for (int n = 0; n < NMAX; ++n) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc[n] += __shfl_xor(acc[n], o);
  }
  if (lane == 0)
    for (int n = 0; n < N; ++n) {
      float v = acc[n] * alpha * (scale ? scale[n] : 1.f) + (shift ? shift[n] : 0.f);
      if (relu) v = fmaxf(v, 0.f);
      c[m * ldc + n] = v;
    }
}

// fp32 [n][k] (row stride ldw) -> three bf16 planes per K-step [kp / 16][3][n][16], kp = k rounded up to 16 (zero padded): the exact split
// of igemm_split_kernel's staging path (h = top 16 bits, m = top 16 bits of x - h, l = x - h - m), done once per weight
// version instead of once per K-step and tile. One lane = four consecutive k of one row.
__global__ void __launch_bounds__(256)
split_weight_kernel(const float* __restrict__ w, long ldw, long batch_w, int n, int k, int kp,
                    unsigned short* __restrict__ out) {
  const int kq = kp / 4;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)n * kq) return;
  const int row = (int)(idx / kq), k4 = (int)(idx - (long)row * kq) * 4;
  const float* src = w + blockIdx.y * batch_w + (long)row * ldw + k4;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (k4 + 3 < k) v = *(const float4*)src;
  else {
    if (k4 < k) v.x = src[0];
    if (k4 + 1 < k) v.y = src[1];
    if (k4 + 2 < k) v.z = src[2];
  }
  uint2 h, m, l;
  split3(v, h, m, l);
  // K-step-major blocks: [kp / 16][3 planes][n][16 bf16] -- the 32 bytes of one (K-step, plane, row) are contiguous and so
  // are the rows of a tile: a wave's staging load of the kernel reads 1 KB of consecutive bytes
  const int ks = k4 / SBK, kk = k4 - ks * SBK;
  unsigned short* o = out + blockIdx.y * 3 * (long)n * kp + (((long)ks * 3) * n + row) * SBK + kk;
  *(uint2*)o = h;
  *(uint2*)(o + (long)n * SBK) = m;
  *(uint2*)(o + 2 * (long)n * SBK) = l;
}

// Epilogue form of the split kernel (dana_set_epilogue_mode): 0 = on the accumulator registers (default), 1 = LDS C tile.
std::atomic<int>& epilogue_mode_cell() {
  static std::atomic<int> cell(0);
  return cell;
}

template <int BM, int BN, int STEM>
int launch(const IgemmParams& p0, int batch, hipStream_t s) {
  IgemmParams p = p0;
  p.tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = (p.N + BN - 1) / BN;
  size_t lds = (size_t)2 * (BM + BN) * LDS_LD * sizeof(float);
  const size_t lds_c = (size_t)BM * (BN + 4) * sizeof(float);
  if (lds_c > lds) lds = lds_c;
  static DeviceOnce attr;  // >64 KiB of dynamic LDS needs the opt-in once per device
  attr.once([&] { return hipFuncSetAttribute((const void*)igemm_f32_kernel<BM, BN, STEM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); });
  dim3 grid(p.tiles_m * p.tiles_n, 1, batch);
  igemm_f32_kernel<BM, BN, STEM><<<grid, 256, lds, s>>>(p);
  return 0;
}

template <int BM, int BN, int STEM = 0, int BPRE = 0, int FUSE = 0, int ELDS = 0>
int launch_split(const IgemmParams& p0, int batch, hipStream_t s) {
  if constexpr (BPRE == 0 && FUSE == 0) {
    if (p0.bpre) return launch_split<BM, BN, STEM, 1, 0, ELDS>(p0, batch, s);
  }
  if constexpr (ELDS == 0 && FUSE == 0) {
    if (epilogue_mode_cell().load(std::memory_order_relaxed)) return launch_split<BM, BN, STEM, BPRE, 0, 1>(p0, batch, s);
  }
  IgemmParams p = p0;
  p.tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = (p.N + BN - 1) / BN;
  size_t lds = (size_t)2 * 3 * (BM + BN) * SLD * sizeof(unsigned);
  const size_t lds_c = (size_t)BM * (BN + 4) * sizeof(float);
  if (lds_c > lds && ELDS) lds = lds_c;
  if (FUSE && lds < (size_t)4 * 3 * BM * 16 * 2) lds = (size_t)4 * 3 * BM * 16 * 2;  // the tile as four K-steps of A
  dim3 grid(p.tiles_m * p.tiles_n, 1, batch);
  static DeviceOnce attr;
  if constexpr (BM * BN >= 128 * 128) {
    attr.once([&] { return hipFuncSetAttribute((const void*)igemm_split_kernel_128<STEM, BPRE, ELDS>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds); });
    igemm_split_kernel_128<STEM, BPRE, ELDS><<<grid, 256, lds, s>>>(p);
  } else {
    attr.once([&] { return hipFuncSetAttribute((const void*)igemm_split_kernel<BM, BN, STEM, BPRE, FUSE, ELDS>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); });
    igemm_split_kernel<BM, BN, STEM, BPRE, FUSE, ELDS><<<grid, 256, lds, s>>>(p);
  }
  return 0;
}

template <int BN, int NST, int APRE, int DF = 0>
int launch_dma(const IgemmParams& p0, int batch, hipStream_t s) {
  IgemmParams p = p0;
  p.tiles_m = (p.M + 127) / 128;
  p.tiles_n = (p.N + BN - 1) / BN;
  const size_t lds = (size_t)NST * 3 * (128 + BN) * 32;
  static DeviceOnce attr;
  attr.once([&] { return hipFuncSetAttribute((const void*)igemm_dma_kernel<BN, NST, APRE, DF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); });
  dim3 grid(p.tiles_m * p.tiles_n, 1, batch);
  igemm_dma_kernel<BN, NST, APRE, DF><<<grid, 256, lds, s>>>(p);
  return 0;
}

// Tile choice (measured on MI355X, tools/conv_sweep.py): the 64x64 block wins or ties on every layer of
// the path -- 36.9 KB of LDS and 68 VGPRs let 4 blocks (16 waves) share a CU, which hides the
// barrier/staging bubbles of the 64-cycle f32 MFMA better than 2 blocks of 128x128, and its finer
// granularity shortens the tail (a CU finishes ceil(tiles/256) tiles while the average is tiles/256).
// DANA_IGEMM_TILE=1|2|3 forces 128x128 | 128x64 | 64x64 for tuning.
// Process-wide kernel choice (every device, every thread: nn.DataParallel's replica threads all see the same value).
// An atomic that is initialised ONCE from DANA_MFMA_SPLIT (thread-safe function-local static) and read ONCE per call:
// a call never mixes modes, concurrent callers never race on the first read. dana_set_mfma_mode is a configuration
// call -- issue it between forwards, not while other threads are inside one.
std::atomic<int>& mfma_mode_cell() {
  static std::atomic<int> cell(getenv("DANA_MFMA_SPLIT") ? atoi(getenv("DANA_MFMA_SPLIT")) : 1);
  return cell;
}
// An A/B knob from the environment, read ONCE per process (function-local statics at the call sites) and validated: an
// unknown value is reported and ignored instead of silently meaning something else. -1 = not set.
int env_choice(const char* name, std::initializer_list<int> allowed) {
  const char* e = getenv(name);
  if (!e || !*e) return -1;
  const int v = atoi(e);
  for (int a : allowed)
    if (a == v) return v;
  fprintf(stderr, "libdana_hip: %s=%s is not one of the accepted values; ignored\n", name, e);
  return -1;
}
unsigned long long* g_trace = nullptr;  // debug: per-block timestamps of the next split launches (dana_set_igemm_trace)

int dispatch(const IgemmParams& p, int batch, int stem, hipStream_t s) {
  if (stem) {
    const int smode = mfma_mode_cell().load(std::memory_order_relaxed);
    return smode ? launch_split<128, 64, 1>(p, batch, s) : launch<128, 64, 1>(p, batch, s);
  }
  // fp32 contractions run on the bf16 matrix cores by default (exact 3-way split, 6 products: igemm_split_kernel);
  // dana_set_mfma_mode(0) / DANA_MFMA_SPLIT=0 selects the f32-MFMA kernel. 128x128 blocks amortise the split best; a
  // launch that cannot give most CUs one of those falls back to 64x64 blocks (measured: tools/conv_sweep.py).
  const int mode = mfma_mode_cell().load(std::memory_order_relaxed);
  if (p.apre) {
    // planes x planes. DANA_PP_STAGES forces a form (tools/pp_probe.py, tests): 2 / 3 / 4 / 6 stages, 13 / 14 = 3 / 4 stages
    // with two fragment sets. Default (profiles/r5_dma_kernel_probe.md): many tiles and a short K walk -> two stages, three
    // workgroups per CU; tile-starved or long-K launches -> three stages, two fragment sets.
    static const int pp_env = env_choice("DANA_PP_STAGES", {2, 3, 4, 6, 13, 14});
    const long t128 = (long)((p.M + 127) / 128) * ((p.N + 127) / 128) * batch;
    const int stages = pp_env >= 0 ? pp_env : ((t128 < 400 || p.K >= 1024) ? 13 : 2);
    if (p.N <= 64) return stages == 2 ? launch_dma<64, 2, 1>(p, batch, s) : launch_dma<64, 3, 1>(p, batch, s);
    if (stages == 2) return launch_dma<128, 2, 1>(p, batch, s);
    if (stages == 4) return launch_dma<128, 4, 1>(p, batch, s);
    if (stages == 6) return launch_dma<128, 6, 1>(p, batch, s);
    if (stages == 13) return launch_dma<128, 3, 1, 1>(p, batch, s);
    if (stages == 14) return launch_dma<128, 4, 1, 1>(p, batch, s);
    return launch_dma<128, 3, 1>(p, batch, s);
  }
  // pre-split weights, no ReLU-adjoint mask: the three-workgroups-per-CU kernel for the launches that took 128 x 128 tiles
  // (DANA_DMA_KERNEL=0: the round-4 kernel, for A/Bs; 2: wherever it can run, 64-wide tiles for N <= 64 -- tests)
  if (mode == 1 && p.bpre && !p.mask && p.KH * p.KW <= 32) {
    static const int dm_env = env_choice("DANA_DMA_KERNEL", {0, 1, 2});
    const int dm = dm_env >= 0 ? dm_env : 1;
    if (dm == 2) return p.N <= 64 ? launch_dma<64, 2, 0>(p, batch, s) : launch_dma<128, 2, 0>(p, batch, s);
    if (dm == 1 && p.N > 64) {
      // measured (profiles/r5_dma_kernel_probe.md, each launch alone): with fp32 activation rows the third workgroup per CU
      // pays on short K walks over many tiles (K = 64 / 128: -4 .. -10 %), is even at K = 256 and loses where a launch is
      // tile-starved or K is long (one staging register set, the split in front of the step's MFMAs: +10 .. +30 %)
      const long t128 = (long)((p.M + 127) / 128) * ((p.N + 127) / 128) * batch;
      if (t128 >= 600 && p.K <= 192) return launch_dma<128, 2, 0>(p, batch, s);
    }
  }
  if (mode && p.KH * p.KW <= 32) {
    if (mode == 2) return launch_split<128, 64>(p, batch, s);
    if (mode == 3) return launch_split<64, 64>(p, batch, s);
    if (mode == 4) return launch_split<128, 128>(p, batch, s);
    if (mode == 5) return launch_split<64, 128>(p, batch, s);
    if (p.N <= 64) return launch_split<128, 64>(p, batch, s);
    // a narrow last column tile (N = 147: 128 + 19) wastes most of a 128-wide tile: 64-wide ones pad less
    if (mode == 1 && p.N <= 256 && p.N % 128 != 0 && p.N % 128 <= 32) return launch_split<64, 64>(p, batch, s);
    const long t128 = (long)((p.M + 127) / 128) * ((p.N + 127) / 128) * batch;
    // Tile choice, re-measured with the round-3 kernel (unpadded LDS: a 128x64 tile runs three workgroups per CU, a 64x64
    // tile four; tools/tile_sweep.py). Launch by launch, ALONE on the chip, smaller tiles win more often now: fewer than
    // ~230 128x128 tiles -> 64x64 (layer3's reduce convs, the Q / K projections: -5 %), a short K walk with one to two
    // rounds of tiles or a residual epilogue behind K <= 256 -> 128x64 (layer2 / layer3 expand convs -7 %); that rule set
    // (removed in round 5) was worth -1 % of the summed launch durations (roofline 0.4175 -> 0.4215) -- and COST 3 % of the
    // step's wall clock (639-643 -> 619-621 query-images/s, same box): the step runs two kernel streams, and many small
    // workgroups of one kernel crowd out the other stream's blocks that would have filled its tail. The default keeps
    // 128x128 wherever the grid can give ~100 CUs one tile.
    if (mode == 1 && t128 < 100) return launch_split<64, 64>(p, batch, s);
    return launch_split<128, 128>(p, batch, s);
  }
  return launch<64, 64, 0>(p, batch, s);
}

int run(IgemmParams& p, int batch, int stem, hipStream_t s) {
  if (p.M0 <= 0 || p.M0 > p.M) p.M0 = p.M;  // single segment
  if (p.M0 == p.M) {
    p.IH1 = p.IH; p.IW1 = p.IW; p.OH1 = p.OH; p.OW1 = p.OW; p.pix1 = 0;
    p.C1 = p.C; p.ldc1 = p.ldc; p.residual1 = p.residual; p.ldr1 = p.ldr;
  }
  p.trace = g_trace;
  p.vec_io = (p.ldc % 4 == 0) && (((uintptr_t)p.C & 15) == 0) && (p.batch_c % 4 == 0) &&
             (!p.residual || ((p.ldr % 4 == 0) && (((uintptr_t)p.residual & 15) == 0))) &&
             (p.ldc1 % 4 == 0) && (((uintptr_t)p.C1 & 15) == 0) &&
             (!p.residual1 || ((p.ldr1 % 4 == 0) && (((uintptr_t)p.residual1 & 15) == 0)));
  p.vec_ss = (((uintptr_t)p.scale | (uintptr_t)p.shift) & 15) == 0;
  return dispatch(p, batch, stem, s);
}

}  // namespace

extern "C" {

int dana_set_epilogue_mode(int mode) {
  DANA_CHECK_ARG(mode == 0 || mode == 1, "dana_set_epilogue_mode: 0 (accumulator registers) or 1 (LDS C tile)");
  epilogue_mode_cell().store(mode, std::memory_order_relaxed);
  return DANA_OK;
}
int dana_get_epilogue_mode(void) { return epilogue_mode_cell().load(std::memory_order_relaxed); }

int dana_set_mfma_mode(int mode) {
  DANA_CHECK_ARG(mode == 0 || mode == 1, "dana_set_mfma_mode: mode must be 0 (f32 MFMA) or 1 (bf16x6 split)");
  mfma_mode_cell().store(mode, std::memory_order_relaxed);
  return DANA_OK;
}

int dana_debug_force_tile(int tile) {
  DANA_CHECK_ARG(tile == 0 || (tile >= 2 && tile <= 5), "dana_debug_force_tile: 0 (dispatcher), 2 = 128x64, 3 = 64x64, 4 = 128x128, 5 = 64x128");
  DANA_CHECK_ARG(mfma_mode_cell().load(std::memory_order_relaxed) != 0, "dana_debug_force_tile: split kernel only");
  mfma_mode_cell().store(tile ? tile : 1, std::memory_order_relaxed);
  return DANA_OK;
}

int dana_set_igemm_trace(unsigned long long* buffer) {
  g_trace = buffer;
  return DANA_OK;
}

int dana_get_mfma_mode(void) {
  return mfma_mode_cell().load(std::memory_order_relaxed) ? 1 : 0;
}

static int conv2d_impl(const char* who, const float* input, const float* weight, float* out0, float* out1,
                       const float* scale, const float* shift, const float* res0, const float* res1, int batch0,
                       int h0, int w0, int batch1, int h1, int w1, int cin, int cout, int kh, int kw, int stride,
                       int pad, long in_pix_stride, long out0_stride, long out1_stride, long res0_stride,
                       long res1_stride, int flags, dana_stream_t stream, const float* mask = nullptr,
                       long mask_stride = 0) {
  DANA_CHECK_ARG(batch0 >= 0 && batch1 >= 0 && cin > 0 && cout > 0 && kh > 0 && kw > 0 && stride > 0 && pad >= 0,
                 "%s: bad shape", who);
  DANA_CHECK_ARG((batch0 == 0 || (h0 > 0 && w0 > 0)) && (batch1 == 0 || (h1 > 0 && w1 > 0)), "%s: bad image size", who);
  if (batch0 + batch1 == 0) return DANA_OK;
  if (batch0 == 0) {  // only the second segment is populated: make it the first
    return conv2d_impl(who, input, weight, out1, nullptr, scale, shift, res1, nullptr, batch1, h1, w1, 0, 0, 0, cin,
                       cout, kh, kw, stride, pad, in_pix_stride, out1_stride, 0, res1_stride, 0, flags, stream);
  }
  DANA_CHECK_ARG(input && weight && out0 && (batch1 == 0 || out1), "%s: null pointer", who);
  DANA_CHECK_ARG(kh * kw <= 64, "%s: at most 64 filter taps", who);
  DANA_CHECK_ARG(!res0 == !(batch1 ? res1 : res0), "%s: residual must be given for both segments or none", who);
  const bool stem = (flags & DANA_CONV_STEM7) != 0;
  IgemmParams p;
  memset(&p, 0, sizeof(p));
  p.A = input;
  p.Bw = weight;
  p.C = out0;
  p.scale = scale;
  p.shift = shift;
  p.residual = res0;
  p.IH = h0;
  p.IW = w0;
  p.OH = (h0 + 2 * pad - kh) / stride + 1;
  p.OW = (w0 + 2 * pad - kw) / stride + 1;
  DANA_CHECK_ARG(p.OH > 0 && p.OW > 0, "%s: empty output", who);
  p.M0 = batch0 * p.OH * p.OW;
  p.M = p.M0;
  if (batch1 > 0) {
    p.IH1 = h1;
    p.IW1 = w1;
    p.OH1 = (h1 + 2 * pad - kh) / stride + 1;
    p.OW1 = (w1 + 2 * pad - kw) / stride + 1;
    DANA_CHECK_ARG(p.OH1 > 0 && p.OW1 > 0, "%s: empty output (segment 1)", who);
    p.pix1 = batch0 * h0 * w0;
    p.M += batch1 * p.OH1 * p.OW1;
    p.C1 = out1;
    p.residual1 = res1;
  }
  p.N = cout;
  p.KH = kh;
  p.KW = kw;
  p.stride = stride;
  p.pad = pad;
  p.alpha = 1.f;
  p.relu = (flags & DANA_EPI_RELU) ? 1 : 0;
  p.mask = mask;
  p.ldm = mask_stride > 0 ? mask_stride : cout;
  DANA_CHECK_ARG(!mask || (batch1 == 0 && p.ldm % 4 == 0 && ((uintptr_t)mask & 15) == 0),
                 "%s: mask needs a single segment and 16-byte aligned rows", who);
  long lda;
  if (stem) {
    // input is NHWC4 (3 channels + zero pad), weight packed [cout][7][8][4] (K = 224)
    DANA_CHECK_ARG(kh == 7 && kw == 7 && cin == 4, "%s: STEM7 needs 7x7 over NHWC4", who);
    p.Cin = 4;
    p.K = 7 * 32;
    lda = 4;
  } else {
    DANA_CHECK_ARG(cin % BK == 0, "%s: cin=%d must be a multiple of %d", who, cin, BK);
    p.Cin = cin;
    p.K = kh * kw * cin;
    lda = in_pix_stride > 0 ? in_pix_stride : cin;
    DANA_CHECK_ARG(lda % 4 == 0 && lda >= cin, "%s: bad in_pix_stride", who);
  }
  p.bpre = (flags & DANA_W_SPLIT3) ? 1 : 0;
  DANA_CHECK_ARG(!p.bpre || (dana_get_mfma_mode() != 0 && (stem || kh * kw <= 32)),
                 "%s: DANA_W_SPLIT3 weights need the split kernel (dana_set_mfma_mode != 0, at most 32 taps)", who);
  const int kpad = (p.K + SBK - 1) / SBK * SBK;
  const long a_bytes = ((long)batch0 * h0 * w0 + (long)batch1 * h1 * w1) * lda * 4;
  const long b_bytes = p.bpre ? (long)3 * cout * kpad * 2 : (long)cout * p.K * 4;
  DANA_CHECK_ARG(a_bytes < (long)OOB && b_bytes < (long)OOB,
                 "%s: operand spans >= 2 GiB are not addressable by one buffer descriptor; split the batch", who);
  p.lda = (int)lda;
  p.ldb = p.bpre ? kpad : p.K;
  p.a_bytes = (unsigned)a_bytes;
  p.b_bytes = (unsigned)b_bytes;
  p.ldc = out0_stride > 0 ? out0_stride : cout;
  p.ldr = res0_stride > 0 ? res0_stride : cout;
  p.ldc1 = out1_stride > 0 ? out1_stride : cout;
  p.ldr1 = res1_stride > 0 ? res1_stride : cout;
  DANA_CHECK_ARG(((uintptr_t)input & 15) == 0 && ((uintptr_t)weight & 15) == 0,
                 "%s: input/weight must be 16-byte aligned", who);
  run(p, 1, stem, (hipStream_t)stream);
  DANA_CHECK_LAUNCH(who);
  return DANA_OK;
}

int dana_conv2d_nhwc(const float* input, const float* weight, float* output, const float* scale,
                     const float* shift, const float* residual, int batch, int in_h, int in_w, int cin,
                     int cout, int kh, int kw, int stride, int pad, long in_pix_stride, long out_pix_stride,
                     long res_pix_stride, int flags, dana_stream_t stream) {
  return conv2d_impl("dana_conv2d_nhwc", input, weight, output, nullptr, scale, shift, residual, nullptr, batch, in_h,
                     in_w, 0, 0, 0, cin, cout, kh, kw, stride, pad, in_pix_stride, out_pix_stride, 0, res_pix_stride,
                     0, flags, stream);
}

int dana_conv2d_nhwc_masked(const float* input, const float* weight, float* output, const float* scale,
                            const float* shift, const float* residual, const float* mask_act, int batch, int in_h,
                            int in_w, int cin, int cout, int kh, int kw, int stride, int pad, long in_pix_stride,
                            long out_pix_stride, long res_pix_stride, long mask_pix_stride, int flags,
                            dana_stream_t stream) {
  return conv2d_impl("dana_conv2d_nhwc_masked", input, weight, output, nullptr, scale, shift, residual, nullptr, batch,
                     in_h, in_w, 0, 0, 0, cin, cout, kh, kw, stride, pad, in_pix_stride, out_pix_stride, 0,
                     res_pix_stride, 0, flags, stream, mask_act, mask_pix_stride);
}

int dana_bottleneck_tail_nhwc(const float* input, const float* w2_split, const float* scale2, const float* shift2,
                              const float* w3_split, const float* scale3, const float* shift3, const float* residual,
                              float* output, int batch, int h, int w, int cin, int cmid, int cout, long in_pix_stride,
                              long out_pix_stride, long res_pix_stride, int flags, dana_stream_t stream) {
  const char* who = "dana_bottleneck_tail_nhwc";
  DANA_CHECK_ARG(batch >= 0 && h > 0 && w > 0, "%s: bad shape", who);
  if (batch == 0) return DANA_OK;
  DANA_CHECK_ARG(input && w2_split && w3_split && output, "%s: null pointer", who);
  DANA_CHECK_ARG(cmid == 64 && cin % SBK == 0 && cin > 0 && cout % 64 == 0 && cout > 0,
                 "%s: needs 64 middle channels, cin %% 16 == 0, cout %% 64 == 0", who);
  DANA_CHECK_ARG(dana_get_mfma_mode() != 0, "%s: split kernel only (dana_set_mfma_mode != 0)", who);
  IgemmParams p;
  memset(&p, 0, sizeof(p));
  p.A = input;
  p.Bw = w2_split;
  p.C = output;
  p.scale = scale2;
  p.shift = shift2;
  p.residual = residual;
  p.IH = p.IH1 = h;
  p.IW = p.IW1 = w;
  p.OH = p.OH1 = h;
  p.OW = p.OW1 = w;
  p.M = p.M0 = batch * h * w;
  p.N = cmid;
  p.KH = p.KW = 3;
  p.stride = 1;
  p.pad = 1;
  p.Cin = cin;
  p.K = 9 * cin;
  p.alpha = 1.f;
  p.relu = 1;
  p.relu2 = (flags & DANA_EPI_RELU) ? 1 : 0;
  p.bpre = 1;
  p.Bw2 = w3_split;
  p.scale2 = scale3;
  p.shift2 = shift3;
  p.N2 = cout;
  const long lda = in_pix_stride > 0 ? in_pix_stride : cin;
  DANA_CHECK_ARG(lda % 4 == 0 && lda >= cin, "%s: bad in_pix_stride", who);
  const long a_bytes = (long)batch * h * w * lda * 4, b_bytes = (long)3 * cmid * p.K * 2;
  DANA_CHECK_ARG(a_bytes < (long)OOB, "%s: input span >= 2 GiB; split the batch", who);
  p.lda = (int)lda;
  p.ldb = p.K;
  p.a_bytes = (unsigned)a_bytes;
  p.b_bytes = (unsigned)b_bytes;
  p.ldc = out_pix_stride > 0 ? out_pix_stride : cout;
  p.ldr = res_pix_stride > 0 ? res_pix_stride : cout;
  p.C1 = p.C;
  p.ldc1 = p.ldc;
  p.residual1 = p.residual;
  p.ldr1 = p.ldr;
  DANA_CHECK_ARG(((uintptr_t)input & 15) == 0 && ((uintptr_t)w2_split & 15) == 0 && ((uintptr_t)w3_split & 15) == 0,
                 "%s: input / weights must be 16-byte aligned", who);
  p.trace = nullptr;
  launch_split<128, 64, 0, 1, 1>(p, 1, (hipStream_t)stream);
  DANA_CHECK_LAUNCH(who);
  return DANA_OK;
}

int dana_conv2d_nhwc_dual(const float* input, const float* weight, float* out0, float* out1, const float* scale,
                          const float* shift, const float* res0, const float* res1, int batch0, int h0, int w0,
                          int batch1, int h1, int w1, int cin, int cout, int kh, int kw, int stride, int pad,
                          long in_pix_stride, long out0_stride, long out1_stride, long res0_stride,
                          long res1_stride, int flags, dana_stream_t stream) {
  return conv2d_impl("dana_conv2d_nhwc_dual", input, weight, out0, out1, scale, shift, res0, res1, batch0, h0, w0,
                     batch1, h1, w1, cin, cout, kh, kw, stride, pad, in_pix_stride, out0_stride, out1_stride,
                     res0_stride, res1_stride, flags, stream);
}

static int cat2_impl(const char* who, const float* a0, long a0_pix_stride, int k0, const float* a1, long a1_pix_stride,
                     int k1, int n0, int h0, int w0, int n1, int h1, int w1, int stride1, const float* weight, float* out0,
                     float* out1, const float* scale, const float* shift, const float* residual, long out0_pix_stride,
                     long out1_pix_stride, long res_pix_stride, int cout, int flags, dana_stream_t stream) {
  DANA_CHECK_ARG(n0 >= 0 && n1 >= 0 && stride1 > 0 && k0 > 0 && k1 > 0 && cout > 0, "%s: bad shape", who);
  DANA_CHECK_ARG((n0 == 0 || (h0 > 0 && w0 > 0)) && (n1 == 0 || (h1 > 0 && w1 > 0)), "%s: bad image size", who);
  if (n0 + n1 == 0) return DANA_OK;
  if (n0 == 0)  // only the second group is populated: make it the first
    return cat2_impl(who, a0, a0_pix_stride, k0, a1, a1_pix_stride, k1, n1, h1, w1, 0, 0, 0, stride1, weight, out1, nullptr,
                     scale, shift, residual, out1_pix_stride, 0, res_pix_stride, cout, flags, stream);
  DANA_CHECK_ARG(a0 && a1 && weight && out0 && (n1 == 0 || out1), "%s: null pointer", who);
  DANA_CHECK_ARG(!(residual && n1), "%s: residual only with one image group", who);
  DANA_CHECK_ARG(k0 % SBK == 0 && k1 % SBK == 0, "%s: k0 and k1 must be multiples of %d", who, SBK);
  DANA_CHECK_ARG(dana_get_mfma_mode() != 0, "%s: needs the split kernel (dana_set_mfma_mode != 0)", who);
  const int oh0 = (h0 - 1) / stride1 + 1, ow0 = (w0 - 1) / stride1 + 1;
  const int oh1 = n1 ? (h1 - 1) / stride1 + 1 : 0, ow1 = n1 ? (w1 - 1) / stride1 + 1 : 0;
  const long lda0 = a0_pix_stride > 0 ? a0_pix_stride : k0, lda1 = a1_pix_stride > 0 ? a1_pix_stride : k1;
  DANA_CHECK_ARG(lda0 % 4 == 0 && lda0 >= k0 && lda1 % 4 == 0 && lda1 >= k1, "%s: bad pixel stride", who);
  const long m0 = (long)n0 * oh0 * ow0, m1 = (long)n1 * oh1 * ow1;
  const long a0_bytes = (m0 + m1) * lda0 * 4, a1_bytes = ((long)n0 * h0 * w0 + (long)n1 * h1 * w1) * lda1 * 4;
  const int bpre = (flags & DANA_W_SPLIT3) ? 1 : 0;  // (k0 + k1 is a multiple of 16: no padding)
  const long b_bytes = bpre ? (long)3 * cout * (k0 + k1) * 2 : (long)cout * (k0 + k1) * 4;
  DANA_CHECK_ARG(a0_bytes < (long)OOB && a1_bytes < (long)OOB && b_bytes < (long)OOB, "%s: operand spans >= 2 GiB", who);
  DANA_CHECK_ARG(((uintptr_t)a0 & 15) == 0 && ((uintptr_t)a1 & 15) == 0 && ((uintptr_t)weight & 15) == 0,
                 "%s: operands must be 16-byte aligned", who);
  IgemmParams p;
  memset(&p, 0, sizeof(p));
  p.A = a0;
  p.Bw = weight;
  p.C = out0;
  p.scale = scale;
  p.shift = shift;
  p.residual = residual;
  p.IH = p.OH = oh0;
  p.IW = p.OW = ow0;
  p.M0 = (int)m0;
  p.M = (int)(m0 + m1);
  if (n1) {  // second image group: a0's rows continue at m0, a1's pixels behind group 0's
    p.IH1 = p.OH1 = oh1;
    p.IW1 = p.OW1 = ow1;
    p.pix1 = (int)m0;
    p.C1 = out1;
    p.IH21 = h1;
    p.IW21 = w1;
    p.pix21 = n0 * h0 * w0;
  }
  p.N = cout;
  p.KH = p.KW = 1;
  p.stride = 1;
  p.pad = 0;
  p.Cin = k0;
  p.K = k0 + k1;
  p.lda = (int)lda0;
  p.ldb = p.K;
  p.a_bytes = (unsigned)a0_bytes;
  p.b_bytes = (unsigned)b_bytes;
  p.bpre = bpre;
  p.A2 = a1;
  p.lda2 = (int)lda1;
  p.IH2 = h0;
  p.IW2 = w0;
  p.stride2 = stride1;
  p.K1 = k1;
  p.a2_bytes = (unsigned)a1_bytes;
  p.ldc = out0_pix_stride > 0 ? out0_pix_stride : cout;
  p.ldr = res_pix_stride > 0 ? res_pix_stride : cout;
  p.ldc1 = n1 ? (out1_pix_stride > 0 ? out1_pix_stride : cout) : p.ldc;
  p.ldr1 = p.ldr;
  p.alpha = 1.f;
  p.relu = (flags & DANA_EPI_RELU) ? 1 : 0;
  p.ldm = cout;
  run(p, 1, 0, (hipStream_t)stream);
  DANA_CHECK_LAUNCH(who);
  return DANA_OK;
}

int dana_conv1x1_cat2_nhwc(const float* a0, long a0_pix_stride, int k0, const float* a1, long a1_pix_stride, int k1,
                           int batch, int h1, int w1, int stride1, const float* weight, float* output,
                           const float* scale, const float* shift, const float* residual, long out_pix_stride,
                           long res_pix_stride, int cout, int flags, dana_stream_t stream) {
  return cat2_impl("dana_conv1x1_cat2_nhwc", a0, a0_pix_stride, k0, a1, a1_pix_stride, k1, batch, h1, w1, 0, 0, 0, stride1,
                   weight, output, nullptr, scale, shift, residual, out_pix_stride, 0, res_pix_stride, cout, flags, stream);
}

int dana_conv1x1_cat2_nhwc_dual(const float* a0, long a0_pix_stride, int k0, const float* a1, long a1_pix_stride, int k1,
                                int n0, int h0, int w0, int n1, int h1, int w1, int stride1, const float* weight,
                                float* out0, float* out1, const float* scale, const float* shift, long out0_pix_stride,
                                long out1_pix_stride, int cout, int flags, dana_stream_t stream) {
  return cat2_impl("dana_conv1x1_cat2_nhwc_dual", a0, a0_pix_stride, k0, a1, a1_pix_stride, k1, n0, h0, w0, n1, h1, w1,
                   stride1, weight, out0, out1, scale, shift, nullptr, out0_pix_stride, out1_pix_stride, 0, cout, flags, stream);
}

size_t dana_split_weight_bytes(int n, int k, int batch) {
  if (n <= 0 || k <= 0 || batch <= 0) return 0;
  return (size_t)batch * 3 * n * ((k + SBK - 1) / SBK * SBK) * 2;
}

int dana_split_weight(const float* w, long ldw, int n, int k, int batch, long batch_w, void* out, dana_stream_t stream) {
  DANA_CHECK_ARG(n > 0 && k > 0 && batch > 0 && ldw >= k, "dana_split_weight: bad shape");
  DANA_CHECK_ARG(w && out && ((uintptr_t)w & 15) == 0 && ((uintptr_t)out & 15) == 0 && ldw % 4 == 0 && batch_w % 4 == 0,
                 "dana_split_weight: pointers / strides must be 16-byte aligned");
  const int kp = (k + SBK - 1) / SBK * SBK;
  dim3 grid((unsigned)dana_ceil_div((long)n * (kp / 4), 256), batch);
  split_weight_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(w, ldw, batch_w, n, k, kp, (unsigned short*)out);
  DANA_CHECK_LAUNCH("dana_split_weight");
  return DANA_OK;
}

int dana_gemm_nt(const float* a, const float* b, float* c, const float* scale, const float* shift,
                 const float* residual, int m, int n, int k, long lda, long ldb, long ldc, long ldr, int batch,
                 long batch_a, long batch_b, long batch_c, float alpha, int flags, dana_stream_t stream) {
  DANA_CHECK_ARG(m >= 0 && n >= 0 && k > 0 && batch >= 0, "dana_gemm_nt: bad shape m=%d n=%d k=%d", m, n, k);
  if (m == 0 || n == 0 || batch == 0) return DANA_OK;
  DANA_CHECK_ARG(a && b && c, "dana_gemm_nt: null pointer");
  DANA_CHECK_ARG(k % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0 && batch_a % 4 == 0 && batch_b % 4 == 0,
                 "dana_gemm_nt: k, lda, ldb and batch strides must be multiples of 4 (16-byte rows)");
  const int bpre = (flags & DANA_W_SPLIT3) ? 1 : 0;  // b = bf16 planes [3][n][ldb], ldb (and batch_b) in bf16 elements
  DANA_CHECK_ARG(!bpre || (dana_get_mfma_mode() != 0 && ldb % SBK == 0 && ldb >= k && batch_b % 8 == 0 && n > 8),
                 "dana_gemm_nt: DANA_W_SPLIT3 needs the split kernel, ldb = k rounded up to 16, n > 8");
  const int apre = (flags & DANA_A_SPLIT3) ? 1 : 0;  // a = bf16 planes [lda / 16][3][m][16], lda = k rounded up to 16 (batch_a in bf16 elements)
  DANA_CHECK_ARG(!apre || (bpre && lda % SBK == 0 && lda >= k && batch_a % 8 == 0),
                 "dana_gemm_nt: DANA_A_SPLIT3 needs DANA_W_SPLIT3 too, lda = k rounded up to 16");
  DANA_CHECK_ARG(lda >= k && ldb >= k && ldc >= n, "dana_gemm_nt: leading dimension smaller than the row");
  DANA_CHECK_ARG(((uintptr_t)a & 15) == 0 && ((uintptr_t)b & 15) == 0, "dana_gemm_nt: a/b must be 16-byte aligned");
  DANA_CHECK_ARG(!residual || batch == 1, "dana_gemm_nt: residual only with batch == 1");
  hipStream_t s = (hipStream_t)stream;
  if (n <= 8 && batch == 1 && !residual) {
    gemm_skinny_kernel<8><<<dana_ceil_div(m, 4), 256, 0, s>>>(a, b, c, scale, shift, m, n, k, lda, ldb, ldc, alpha,
                                                              (flags & DANA_EPI_RELU) ? 1 : 0);
    DANA_CHECK_LAUNCH("dana_gemm_nt(skinny)");
    return DANA_OK;
  }
  const long a_bytes = apre ? (long)3 * m * lda * 2 : ((long)(m - 1) * lda + k) * 4;
  const long b_bytes = bpre ? (long)3 * n * ldb * 2 : ((long)(n - 1) * ldb + k) * 4;
  DANA_CHECK_ARG(a_bytes < (long)OOB && b_bytes < (long)OOB, "dana_gemm_nt: operand slice >= 2 GiB; split it");
  IgemmParams p;
  memset(&p, 0, sizeof(p));
  p.A = a;
  p.Bw = b;
  p.C = c;
  p.scale = scale;
  p.shift = shift;
  p.residual = residual;
  p.M = m;
  p.N = n;
  p.K = k;
  // a GEMM is a 1x1 conv over an M x 1 "image" whose channel count covers the whole K range
  p.IH = m;
  p.IW = 1;
  p.OH = m;
  p.OW = 1;
  p.Cin = (k + BK - 1) / BK * BK;
  p.KH = p.KW = 1;
  p.stride = 1;
  p.pad = 0;
  p.lda = (int)lda;
  p.ldb = (int)ldb;
  p.a_bytes = (unsigned)a_bytes;
  p.b_bytes = (unsigned)b_bytes;
  p.ldc = ldc;
  p.ldr = ldr > 0 ? ldr : ldc;
  p.batch_a = apre ? batch_a / 2 : batch_a;
  p.batch_b = bpre ? batch_b / 2 : batch_b;  // (the kernel steps the filter pointer in floats)
  p.apre = apre;
  p.a_rows = m;
  p.batch_c = batch_c;
  p.bpre = bpre;
  p.alpha = alpha;
  p.relu = (flags & DANA_EPI_RELU) ? 1 : 0;
  run(p, batch, 0, s);
  DANA_CHECK_LAUNCH("dana_gemm_nt");
  return DANA_OK;
}

}  // extern "C"
