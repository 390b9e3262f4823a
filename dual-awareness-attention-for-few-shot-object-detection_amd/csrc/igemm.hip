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
// im2col with zero fill for padding), B rows are filters. Epilogue fuses frozen-BN scale/shift or
// bias, residual add and ReLU (resnet.py:84-100), and writes with row stride `ldc` so producers can
// write straight into concat buffers (dana.py:153-154 torch.cat eliminated).
//
// Tiling (wave64, 4 waves as 2x2): block BMxBNx32, wave tile (BM/2)x(BN/2) made of 32x32 MFMA tiles.
// LDS rows are padded to 36 dwords: the ds_read_b128 fragment reads (16-lane groups, distinct rows)
// and the 8-lane ds_write_b128 staging writes are both bank-conflict-free. K is consumed 8 at a
// time per lane-half: lane (i, h) holds k = 8c+4h .. 8c+4h+3 of row i for BOTH operands, so MFMA
// step s multiplies the k-pair {8c+s, 8c+4+s}; any pairing is valid because A and B agree.
// Pipeline: global->register prefetch of tile t+1 is issued before the MFMAs of tile t, written
// to the other LDS buffer after them; one barrier per K-step. The 64-cycle f32 MFMA hides the rest.
#include "common.h"
#include "../../include/dana_hip.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct IgemmParams {
  const float* A;
  const float* Bw;
  float* C;
  const float* scale;     // [N] or null
  const float* shift;     // [N] or null
  const float* residual;  // [M][ldr] or null
  int M, N, K;
  int IH, IW, OH, OW, Cin, KH, KW, stride, pad;
  long lda, ldb, ldc, ldr;
  long batch_a, batch_b, batch_c;  // blockIdx.z strides (floats)
  float alpha;
  int relu;
  int tiles_m, tiles_n;
};

constexpr int BK = 32;
constexpr int LDS_LD = 36;  // padded row, dwords

// bijective XCD-aware remap: consecutive tile ids land on the same XCD's L2 (guide T1)
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg / 8, r = nwg % 8;
  const int xcd = bid % 8, local = bid / 8;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
}

template <int BM, int BN, int STEM>
__global__ void __launch_bounds__(256, 2) igemm_f32_kernel(IgemmParams p) {
  constexpr int TM = BM / 64, TN = BN / 64;  // 32x32 MFMA tiles per wave in m / n
  constexpr int RA = BM / 32, RB = BN / 32;  // float4 loads per thread per K-step
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                       // [2][BM][LDS_LD]
  float* Bs = smem + 2 * BM * LDS_LD;     // [2][BN][LDS_LD]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;

  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int tm_idx = tile / p.tiles_n, tn_idx = tile % p.tiles_n;
  const int m0 = tm_idx * BM, n0 = tn_idx * BN;
  const float* Ab = p.A + (long)blockIdx.z * p.batch_a;
  const float* Bb = p.Bw + (long)blockIdx.z * p.batch_b;
  float* Cb = p.C + (long)blockIdx.z * p.batch_c;

  // ---- per-thread staging coordinates --------------------------------------------------------
  const int c4 = tid & 7;    // which float4 of the 32-wide k chunk
  const int r0 = tid >> 3;   // row within a 32-row slab
  long a_pix[RA];            // pixel index of (img, 0, 0) for this row
  int a_ih0[RA], a_iw0[RA];
  bool a_ok[RA];
#pragma unroll
  for (int j = 0; j < RA; ++j) {
    const int m = m0 + r0 + 32 * j;
    a_ok[j] = m < p.M;
    const int mm = a_ok[j] ? m : 0;
    const int ohw = p.OH * p.OW;
    const int img = mm / ohw, rem = mm % ohw;
    const int oh = rem / p.OW, ow = rem % p.OW;
    a_pix[j] = (long)img * p.IH * p.IW;
    a_ih0[j] = oh * p.stride - p.pad;
    a_iw0[j] = ow * p.stride - p.pad;
  }
  const float* b_ptr[RB];
  bool b_ok[RB];
#pragma unroll
  for (int j = 0; j < RB; ++j) {
    const int n = n0 + r0 + 32 * j;
    b_ok[j] = n < p.N;
    b_ptr[j] = Bb + (long)(b_ok[j] ? n : 0) * p.ldb + c4 * 4;
  }

  float4 ra[RA], rb[RB];
  auto load_tile = [&](int kt) {
    const int k0 = kt * BK;
    const bool kok = (k0 + c4 * 4) < p.K;
    if (STEM) {
      // chunk kt = filter row kh; float4 c4 = tap kw (8th tap and 4th channel are zero weights)
#pragma unroll
      for (int j = 0; j < RA; ++j) {
        const int ih = a_ih0[j] + kt, iw = a_iw0[j] + c4;
        const bool ok = a_ok[j] && c4 < 7 && ih >= 0 && ih < p.IH && iw >= 0 && iw < p.IW;
        ra[j] = ok ? *(const float4*)(Ab + (a_pix[j] + (long)ih * p.IW + iw) * 4) : make_float4(0, 0, 0, 0);
      }
    } else {
      const int tap = k0 / p.Cin, cin0 = k0 - tap * p.Cin;
      const int kh = tap / p.KW, kw = tap - kh * p.KW;
#pragma unroll
      for (int j = 0; j < RA; ++j) {
        const int ih = a_ih0[j] + kh, iw = a_iw0[j] + kw;
        const bool ok = a_ok[j] && kok && ih >= 0 && ih < p.IH && iw >= 0 && iw < p.IW;
        ra[j] = ok ? *(const float4*)(Ab + (a_pix[j] + (long)ih * p.IW + iw) * p.lda + cin0 + c4 * 4)
                   : make_float4(0, 0, 0, 0);
      }
    }
#pragma unroll
    for (int j = 0; j < RB; ++j)
      rb[j] = (b_ok[j] && kok) ? *(const float4*)(b_ptr[j] + k0) : make_float4(0, 0, 0, 0);
  };
  auto store_tile = [&](int buf) {
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

  const int nk = (p.K + BK - 1) / BK;
  load_tile(0);
  store_tile(0);
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) load_tile(kt + 1);  // global loads in flight across the MFMAs below
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
    if (kt + 1 < nk) store_tile(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: C/D map col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5) -------------------
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + wn * (BN / 2) + j * 32 + li;
    const bool nok = n < p.N;
    const float sc = (nok && p.scale) ? p.scale[n] : 1.f;
    const float sh = (nok && p.shift) ? p.shift[n] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int mb = m0 + wm * (BM / 2) + i * 32 + 4 * lh;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mb + (r & 3) + 8 * (r >> 2);
        if (nok && m < p.M) {
          float v = acc[i][j][r] * p.alpha;
          v = v * sc + sh;
          if (p.residual) v += p.residual[(long)m * p.ldr + n];
          if (p.relu) v = fmaxf(v, 0.f);
          Cb[(long)m * p.ldc + n] = v;
        }
      }
    }
  }
}

template <int BM, int BN, int STEM>
int launch(const IgemmParams& p0, int batch, hipStream_t s) {
  IgemmParams p = p0;
  p.tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = (p.N + BN - 1) / BN;
  const size_t lds = (size_t)2 * (BM + BN) * LDS_LD * sizeof(float);
  static bool attr_set = false;  // >64 KiB of dynamic LDS needs the opt-in once per process
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)igemm_f32_kernel<BM, BN, STEM>, hipFuncAttributeMaxDynamicSharedMemorySize,
                        (int)lds);
    attr_set = true;
  }
  dim3 grid(p.tiles_m * p.tiles_n, 1, batch);
  igemm_f32_kernel<BM, BN, STEM><<<grid, 256, lds, s>>>(p);
  return 0;
}

int dispatch(const IgemmParams& p, int batch, int stem, hipStream_t s) {
  if (stem) return launch<128, 64, 1>(p, batch, s);
  // tile choice: keep >= ~2 waves of workgroups over the 256 CUs when the problem allows it
  const long t128 = (long)((p.M + 127) / 128) * ((p.N + 127) / 128) * batch;
  if (p.N <= 64) {
    return launch<128, 64, 0>(p, batch, s);
  }
  if (t128 >= 384) return launch<128, 128, 0>(p, batch, s);
  const long t12864 = (long)((p.M + 127) / 128) * ((p.N + 63) / 64) * batch;
  if (t12864 >= 384) return launch<128, 64, 0>(p, batch, s);
  return launch<64, 64, 0>(p, batch, s);
}

}  // namespace

extern "C" {

int dana_conv2d_nhwc(const float* input, const float* weight, float* output, const float* scale,
                     const float* shift, const float* residual, int batch, int in_h, int in_w, int cin,
                     int cout, int kh, int kw, int stride, int pad, long in_pix_stride, long out_pix_stride,
                     long res_pix_stride, int flags, dana_stream_t stream) {
  DANA_CHECK_ARG(batch >= 0 && in_h > 0 && in_w > 0 && cin > 0 && cout > 0 && kh > 0 && kw > 0 && stride > 0 &&
                     pad >= 0,
                 "dana_conv2d_nhwc: bad shape");
  if (batch == 0) return DANA_OK;
  DANA_CHECK_ARG(input && weight && output, "dana_conv2d_nhwc: null pointer");
  const bool stem = (flags & DANA_CONV_STEM7) != 0;
  IgemmParams p;
  memset(&p, 0, sizeof(p));
  p.A = input;
  p.Bw = weight;
  p.C = output;
  p.scale = scale;
  p.shift = shift;
  p.residual = residual;
  p.IH = in_h;
  p.IW = in_w;
  p.OH = (in_h + 2 * pad - kh) / stride + 1;
  p.OW = (in_w + 2 * pad - kw) / stride + 1;
  DANA_CHECK_ARG(p.OH > 0 && p.OW > 0, "dana_conv2d_nhwc: empty output");
  p.M = batch * p.OH * p.OW;
  p.N = cout;
  p.KH = kh;
  p.KW = kw;
  p.stride = stride;
  p.pad = pad;
  p.alpha = 1.f;
  p.relu = (flags & DANA_EPI_RELU) ? 1 : 0;
  if (stem) {
    // input is NHWC4 (3 channels + zero pad), weight packed [cout][7][8][4] (K = 224)
    DANA_CHECK_ARG(kh == 7 && kw == 7 && cin == 4, "dana_conv2d_nhwc: STEM7 needs 7x7 over NHWC4");
    p.Cin = 4;
    p.K = 7 * 32;
    p.lda = 4;
  } else {
    DANA_CHECK_ARG(cin % BK == 0, "dana_conv2d_nhwc: cin=%d must be a multiple of %d", cin, BK);
    p.Cin = cin;
    p.K = kh * kw * cin;
    p.lda = in_pix_stride > 0 ? in_pix_stride : cin;
    DANA_CHECK_ARG(p.lda % 4 == 0, "dana_conv2d_nhwc: in_pix_stride %% 4 != 0");
  }
  p.ldb = p.K;
  p.ldc = out_pix_stride > 0 ? out_pix_stride : cout;
  p.ldr = res_pix_stride > 0 ? res_pix_stride : cout;
  DANA_CHECK_ARG(((uintptr_t)input & 15) == 0 && ((uintptr_t)weight & 15) == 0,
                 "dana_conv2d_nhwc: input/weight must be 16-byte aligned");
  dispatch(p, 1, stem, (hipStream_t)stream);
  DANA_CHECK_LAUNCH("dana_conv2d_nhwc");
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
  DANA_CHECK_ARG(((uintptr_t)a & 15) == 0 && ((uintptr_t)b & 15) == 0, "dana_gemm_nt: a/b must be 16-byte aligned");
  DANA_CHECK_ARG(!residual || batch == 1, "dana_gemm_nt: residual only with batch == 1");
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
  p.lda = lda;
  p.ldb = ldb;
  p.ldc = ldc;
  p.ldr = ldr > 0 ? ldr : ldc;
  p.batch_a = batch_a;
  p.batch_b = batch_b;
  p.batch_c = batch_c;
  p.alpha = alpha;
  p.relu = (flags & DANA_EPI_RELU) ? 1 : 0;
  dispatch(p, batch, 0, (hipStream_t)stream);
  DANA_CHECK_LAUNCH("dana_gemm_nt");
  return DANA_OK;
}

}  // extern "C"
