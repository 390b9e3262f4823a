// HBM-bound layout / pooling / packing kernels of the DAnA forward path (gfx950).
// All activations are NHWC; every kernel moves 16 B per lane along the channel axis where the
// shape allows it, one pass over its input, grid >> 256 workgroups.
//
// Reference semantics replaced (not code):
//   nn.MaxPool2d(3, 2, padding=0, ceil_mode=True)      lib/model/framework/resnet.py:113
//   nn.AvgPool2d(14, stride=1) on the support maps     lib/model/framework/dana.py:42,105-108
//   .mean(3).mean(2) after layer4                      lib/model/framework/dana.py:387-389
//   PositionalEncoding.forward (x + pe)                lib/model/framework/dana.py:322-324
//   q - q.mean(1, keepdim=True)                        lib/model/framework/dana.py:125,141,267,272
//   frozen BatchNorm2d (eval) folded to scale/shift    lib/model/framework/dana.py:362-385
#include "common.h"
#include "../../include/dana_hip.h"
#include <float.h>

namespace {

int grid_for(long total, int block) {
  long g = (total + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > 65535L * 16 ? 65535L * 16 : g));
}

// [B][C][H][W] -> [B][H][W][ldo], channels >= C zero-filled up to cpad
__global__ void __launch_bounds__(256)
nchw_to_nhwc_kernel(const float* __restrict__ in, float* __restrict__ out, int C, int HW, int cpad, long ldo,
                    long total) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)blockDim.x * gridDim.x) {
    const int c = (int)(i % cpad);
    const long pix = i / cpad;  // b*HW + hw
    const long b = pix / HW, hw = pix % HW;
    out[pix * ldo + c] = c < C ? in[(b * C + c) * HW + hw] : 0.f;
  }
}

// the input images: [B][3][HW] -> [B][HW][4] (4th channel zero): one pixel per lane, three coalesced plane reads, one
// 16-byte store (the general kernel above pays a 64-bit divide per ELEMENT and reads with a 4-lane stride)
__global__ void __launch_bounds__(256)
nchw3_to_nhwc4_kernel(const float* __restrict__ in, float4* __restrict__ out, int HW, int total_pix) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total_pix; i += blockDim.x * gridDim.x) {
    const int b = i / HW, hw = i - b * HW;
    const float* p = in + (long)b * 3 * HW + hw;
    out[i] = make_float4(p[0], p[HW], p[2 * (long)HW], 0.f);
  }
}

// [B][HW][ldi] -> [B][C][HW]
__global__ void __launch_bounds__(256)
nhwc_to_nchw_kernel(const float* __restrict__ in, float* __restrict__ out, int C, int HW, long ldi, long total) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)blockDim.x * gridDim.x) {
    const long hw = i % HW;
    const long c = (i / HW) % C;
    const long b = i / HW / C;
    out[i] = in[(b * HW + hw) * ldi + c];
  }
}

// 3x3 stride-2 max pool, pad 0, ceil_mode: windows are clipped at the bottom/right edge.
__global__ void __launch_bounds__(256)
maxpool3x3s2_kernel(const float4* __restrict__ in, float4* __restrict__ out, int H, int W, int OH, int OW, int C4,
                    long total) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)blockDim.x * gridDim.x) {
    const int c = (int)(i % C4);
    const int ow = (int)((i / C4) % OW);
    const int oh = (int)((i / C4 / OW) % OH);
    const long b = i / C4 / OW / OH;
    float4 m = make_float4(-FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX);
    const int h0 = oh * 2, w0 = ow * 2;
    for (int dh = 0; dh < 3; ++dh) {
      const int h = h0 + dh;
      if (h >= H) break;
      for (int dw = 0; dw < 3; ++dw) {
        const int w = w0 + dw;
        if (w >= W) break;
        const float4 v = in[((b * H + h) * W + w) * C4 + c];
        m.x = fmaxf(m.x, v.x);
        m.y = fmaxf(m.y, v.y);
        m.z = fmaxf(m.z, v.z);
        m.w = fmaxf(m.w, v.w);
      }
    }
    out[i] = m;
  }
}

// nn.MaxPool2d(2): 2x2 stride-2 max pool, floor mode (meta.py:203,246: the PRN's pooling of the support map)
__global__ void __launch_bounds__(256)
maxpool2x2s2_kernel(const float4* __restrict__ in, float4* __restrict__ out, int H, int W, int OH, int OW, int C4,
                    long total) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)blockDim.x * gridDim.x) {
    const int c = (int)(i % C4);
    const int ow = (int)((i / C4) % OW);
    const int oh = (int)((i / C4 / OW) % OH);
    const long b = i / C4 / OW / OH;
    const float4* p = in + ((b * H + oh * 2) * W + ow * 2) * C4 + c;
    const float4 a = p[0], bb = p[C4], cc = p[(long)W * C4], d = p[(long)W * C4 + C4];
    out[i] = make_float4(fmaxf(fmaxf(a.x, bb.x), fmaxf(cc.x, d.x)), fmaxf(fmaxf(a.y, bb.y), fmaxf(cc.y, d.y)),
                         fmaxf(fmaxf(a.z, bb.z), fmaxf(cc.z, d.z)), fmaxf(fmaxf(a.w, bb.w), fmaxf(cc.w, d.w)));
  }
}

// adjoint of nn.MaxPool2d(2): each window's gradient goes to its maximum -- the FIRST one in row-major window order on
// ties, like torch's max_pool2d backward; rows / columns the floor-mode pooling never reads get 0. The argmax is
// recomputed from the forward input.
__global__ void __launch_bounds__(256)
maxpool2x2s2_bwd_kernel(const float* __restrict__ in, const float* __restrict__ gout, float* __restrict__ gin, int H, int W,
                        int OH, int OW, int C, long total) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)blockDim.x * gridDim.x) {
    const int c = (int)(i % C);
    const int x = (int)((i / C) % W);
    const int y = (int)((i / C / W) % H);
    const long b = i / C / W / H;
    const int oh = y >> 1, ow = x >> 1;
    float g = 0.f;
    if (oh < OH && ow < OW) {
      const float* p = in + ((b * H + oh * 2) * W + ow * 2) * C + c;
      const float v[4] = {p[0], p[C], p[(long)W * C], p[(long)W * C + C]};
      int arg = 0;
#pragma unroll
      for (int k = 1; k < 4; ++k)
        if (v[k] > v[arg]) arg = k;
      if (arg == (y & 1) * 2 + (x & 1)) g = gout[((b * OH + oh) * OW + ow) * C + c];
    }
    gin[i] = g;
  }
}

// y = 1 / (1 + exp(-x)), in place (meta.py:202,250)
__global__ void __launch_bounds__(256) sigmoid_kernel(float* __restrict__ x, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)blockDim.x * gridDim.x)
    x[i] = 1.f / (1.f + expf(-x[i]));
}

// out[r][c] = x[r][c] * vec[r / rows_per_group][c]  (meta.py:136-140: RoI features x the class-attentive vector)
__global__ void __launch_bounds__(256)
scale_rows_by_group_kernel(const float4* __restrict__ x, const float4* __restrict__ vec, float4* __restrict__ out,
                           long rows_per_group, int C4, long total) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)blockDim.x * gridDim.x) {
    const int c = (int)(i % C4);
    const long g = (i / C4) / rows_per_group;
    const float4 a = x[i], v = vec[g * C4 + c];
    out[i] = make_float4(a.x * v.x, a.y * v.y, a.z * v.z, a.w * v.w);
  }
}

// depth-wise "valid" cross-correlation (fsod.py:109-116,207-214: F.conv2d(feat, kernel.view(C,1,kh,kw), groups=C)):
// out[n][oh][ow][c] = sum_{i,j} feat[n][oh+i][ow+j][c] * kern[n / per_kernel][i][j][c]
__global__ void __launch_bounds__(256)
depthwise_corr_kernel(const float4* __restrict__ feat, const float4* __restrict__ kern, float4* __restrict__ out, int H,
                      int W, int KH, int KW, int OH, int OW, int C4, long lda4, long per_kernel, long total) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)blockDim.x * gridDim.x) {
    const int c = (int)(i % C4);
    const int ow = (int)((i / C4) % OW);
    const int oh = (int)((i / C4 / OW) % OH);
    const long n = i / C4 / OW / OH;
    const float4* kp = kern + (n / per_kernel) * KH * KW * C4 + c;
    const float4* fp = feat + ((n * H + oh) * W + ow) * lda4 + c;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int a = 0; a < KH; ++a)
      for (int b = 0; b < KW; ++b) {
        const float4 f = fp[((long)a * W + b) * lda4], k = kp[(a * KW + b) * C4];
        acc.x += f.x * k.x;
        acc.y += f.y * k.y;
        acc.z += f.z * k.z;
        acc.w += f.w * k.w;
      }
    out[i] = acc;
  }
}

// adjoints of the depth-wise "valid" cross-correlation above (the fsod sibling's backward):
//   d feat[n][y][x][c]  = sum_{i,j : 0 <= y-i < OH, 0 <= x-j < OW} d out[n][y-i][x-j][c] * kern[n / per_kernel][i][j][c]
//   d kern[k][i][j][c]  = sum_{n in kernel k's maps} sum_{oh,ow} d out[n][oh][ow][c] * feat[n][oh+i][ow+j][c]
__global__ void __launch_bounds__(256)
depthwise_corr_bwd_feat_kernel(const float4* __restrict__ gout, const float4* __restrict__ kern, float4* __restrict__ gfeat,
                               int H, int W, int KH, int KW, int OH, int OW, int C4, long per_kernel, long total) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)blockDim.x * gridDim.x) {
    const int c = (int)(i % C4);
    const int x = (int)((i / C4) % W);
    const int y = (int)((i / C4 / W) % H);
    const long n = i / C4 / W / H;
    const float4* kp = kern + (n / per_kernel) * KH * KW * C4 + c;
    const float4* gp = gout + n * OH * OW * C4 + c;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int a = 0; a < KH; ++a) {
      const int oh = y - a;
      if (oh < 0 || oh >= OH) continue;
      for (int b = 0; b < KW; ++b) {
        const int ow = x - b;
        if (ow < 0 || ow >= OW) continue;
        const float4 g = gp[((long)oh * OW + ow) * C4], k = kp[(a * KW + b) * C4];
        acc.x += g.x * k.x;
        acc.y += g.y * k.y;
        acc.z += g.z * k.z;
        acc.w += g.w * k.w;
      }
    }
    gfeat[i] = acc;
  }
}

// one thread per (kernel, tap, float4 of channels); maps and output positions are walked in order (deterministic)
__global__ void __launch_bounds__(256)
depthwise_corr_bwd_kern_kernel(const float4* __restrict__ gout, const float4* __restrict__ feat, float4* __restrict__ gkern,
                               int H, int W, int KH, int KW, int OH, int OW, int C4, long lda4, long per_kernel,
                               long n_maps, long total, int accumulate) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)blockDim.x * gridDim.x) {
    const int c = (int)(i % C4);
    const int b = (int)((i / C4) % KW);
    const int a = (int)((i / C4 / KW) % KH);
    const long k = i / C4 / KW / KH;
    float4 acc = accumulate ? gkern[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    const long n1 = (k + 1) * per_kernel < n_maps ? (k + 1) * per_kernel : n_maps;
    for (long n = k * per_kernel; n < n1; ++n) {
      const float4* gp = gout + n * OH * OW * C4 + c;
      const float4* fp = feat + ((n * H + a) * W + b) * lda4 + c;
      for (int oh = 0; oh < OH; ++oh)
        for (int ow = 0; ow < OW; ++ow) {
          const float4 g = gp[((long)oh * OW + ow) * C4], f = fp[((long)oh * W + ow) * lda4];
          acc.x += g.x * f.x;
          acc.y += g.y * f.y;
          acc.z += g.z * f.z;
          acc.w += g.w * f.w;
        }
    }
    gkern[i] = acc;
  }
}

// k x k average pool, given stride, no padding (dana.py:42: AvgPool2d(14, stride=1))
__global__ void __launch_bounds__(256)
avgpool_kernel(const float4* __restrict__ in, float4* __restrict__ out, int H, int W, int OH, int OW, int C4, int k,
               int stride, long total) {
  const float cnt = (float)(k * k);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)blockDim.x * gridDim.x) {
    const int c = (int)(i % C4);
    const int ow = (int)((i / C4) % OW);
    const int oh = (int)((i / C4 / OW) % OH);
    const long b = i / C4 / OW / OH;
    float4 s = make_float4(0, 0, 0, 0);
    for (int dh = 0; dh < k; ++dh)
      for (int dw = 0; dw < k; ++dw) {
        const float4 v = in[((b * H + oh * stride + dh) * W + ow * stride + dw) * C4 + c];
        s.x += v.x;
        s.y += v.y;
        s.z += v.z;
        s.w += v.w;
      }
    out[i] = make_float4(s.x / cnt, s.y / cnt, s.z / cnt, s.w / cnt);
  }
}

// stride-1 variant for OW <= 8 (the 20x20 -> 7x7 support pooling): one thread = (image, output row, 4 channels) walks
// its k input rows once and feeds every pixel to the <= 8 windows that contain it, instead of re-reading each pixel
// from k*k windows (PMC: the generic kernel fetched 883 MB for a 39 MB input).
__global__ void __launch_bounds__(256)
avgpool_rows_kernel(const float4* __restrict__ in, float4* __restrict__ out, int H, int W, int OH, int OW, int C4, int k,
                    long total) {
  const float cnt = (float)(k * k);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)blockDim.x * gridDim.x) {
    const int c = (int)(i % C4);
    const int oh = (int)((i / C4) % OH);
    const long b = i / C4 / OH;
    float4 acc[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) acc[o] = make_float4(0, 0, 0, 0);
    for (int dh = 0; dh < k; ++dh) {
      const float4* row = in + ((b * H + oh + dh) * W) * C4 + c;
      for (int x = 0; x < W; ++x) {
        const float4 v = row[(long)x * C4];
#pragma unroll
        for (int o = 0; o < 8; ++o)
          if (o <= x && x < o + k) {
            acc[o].x += v.x;
            acc[o].y += v.y;
            acc[o].z += v.z;
            acc[o].w += v.w;
          }
      }
    }
#pragma unroll
    for (int o = 0; o < 8; ++o)
      if (o < OW) out[((b * OH + oh) * OW + o) * C4 + c] = make_float4(acc[o].x / cnt, acc[o].y / cnt, acc[o].z / cnt, acc[o].w / cnt);
  }
}

// stride-1 pooling of a small map (the 20x20 -> 7x7 support pooling, dana.py:105-108) as a separable box filter through
// LDS: a workgroup owns (image, 16 channels), reads its H x W x 16 slab ONCE (the rows kernel above re-read every input
// row from up to 7 output rows and ran 168 workgroups of 280 dependent loads: 150 us for a 39 MB input), sums the k
// columns of every window into a second LDS image and the k rows of that into the output.
constexpr int AP_CH4 = 4;  // float4s (= 16 channels) per workgroup
__global__ void __launch_bounds__(256)
avgpool_tile_kernel(const float4* __restrict__ in, float4* __restrict__ out, int H, int W, int OH, int OW, int C4, int k) {
  extern __shared__ __attribute__((aligned(16))) float4 ap_lds[];  // [H*W][AP_CH4] then [H*OW][AP_CH4]
  float4* tile = ap_lds;
  float4* hsum = ap_lds + H * W * AP_CH4;
  const int groups = C4 / AP_CH4;
  const long b = blockIdx.x / groups;
  const int c0 = (blockIdx.x % groups) * AP_CH4;
  const int tid = threadIdx.x;
  for (int i = tid; i < H * W * AP_CH4; i += 256) tile[i] = in[(b * H * W + i / AP_CH4) * C4 + c0 + (i % AP_CH4)];
  __syncthreads();
  for (int i = tid; i < H * OW * AP_CH4; i += 256) {
    const int q = i % AP_CH4, ox = (i / AP_CH4) % OW, y = i / AP_CH4 / OW;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int dx = 0; dx < k; ++dx) {
      const float4 v = tile[(y * W + ox + dx) * AP_CH4 + q];
      s.x += v.x;
      s.y += v.y;
      s.z += v.z;
      s.w += v.w;
    }
    hsum[i] = s;
  }
  __syncthreads();
  const float inv = 1.f / (float)(k * k);
  for (int i = tid; i < OH * OW * AP_CH4; i += 256) {
    const int q = i % AP_CH4, ox = (i / AP_CH4) % OW, oy = i / AP_CH4 / OW;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int dy = 0; dy < k; ++dy) {
      const float4 v = hsum[((oy + dy) * OW + ox) * AP_CH4 + q];
      s.x += v.x;
      s.y += v.y;
      s.z += v.z;
      s.w += v.w;
    }
    out[((b * OH + oy) * OW + ox) * C4 + c0 + q] = make_float4(s.x * inv, s.y * inv, s.z * inv, s.w * inv);
  }
}

// out[g][c] = mean_p in[g][p][c]
__global__ void __launch_bounds__(256)
spatial_mean_kernel(const float4* __restrict__ in, float4* __restrict__ out, int P, int C4, long ldi4, long total) {
  const float cnt = (float)P;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)blockDim.x * gridDim.x) {
    const int c = (int)(i % C4);
    const long g = i / C4;
    float4 s = make_float4(0, 0, 0, 0);
    for (int p = 0; p < P; ++p) {
      const float4 v = in[(g * P + p) * ldi4 + c];
      s.x += v.x;
      s.y += v.y;
      s.z += v.z;
      s.w += v.w;
    }
    out[i] = make_float4(s.x / cnt, s.y / cnt, s.z / cnt, s.w / cnt);
  }
}

// out[r][c] = in[r][c] + pe[r % L][c]   (rows = groups * L)
__global__ void __launch_bounds__(256)
add_pe_kernel(const float4* __restrict__ in, const float4* __restrict__ pe, float4* __restrict__ out, int L, int C4,
              long ldi4, long ldo4, long total) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)blockDim.x * gridDim.x) {
    const int c = (int)(i % C4);
    const long r = i / C4;
    const float4 a = in[r * ldi4 + c], b = pe[(r % L) * C4 + c];
    out[r * ldo4 + c] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
  }
}

// out[g][r][c] = in[g * in_group + r * C + c] + pe[r % L][c]: the positive supports of every image in ONE launch (their maps
// sit way * shot images apart in the support batch: dana.py:103,126-130)
__global__ void __launch_bounds__(256)
add_pe_groups_kernel(const float4* __restrict__ in, const float4* __restrict__ pe, float4* __restrict__ out, int L, int C4,
                     long rows_per_group, long in_group4, long out_group4, long total) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)blockDim.x * gridDim.x) {
    const int c = (int)(i % C4);
    const long rr = i / C4;
    const long g = rr / rows_per_group, r = rr % rows_per_group;
    const float4 a = in[g * in_group4 + r * C4 + c], b = pe[(r % L) * C4 + c];
    out[g * out_group4 + r * C4 + c] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
  }
}

// x[g][l][d] -= mean_l x[g][l][d] in two passes over row chunks (fixed summation order, no atomics):
// (1) partial[g][chunk][d] = sum of the chunk's rows; (2) every chunk re-adds the partials in chunk
// order, divides by L and subtracts. grid = (D/64 column slabs, chunks, G); block = 64 cols x 4 row lanes.
constexpr int CM_ROWS = 64;
__global__ void __launch_bounds__(256)
colmean_partial_kernel(const float* __restrict__ x, float* __restrict__ partial, int L, int D, long ld) {
  __shared__ float part[4][64];
  const int col = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rl = threadIdx.x >> 6;
  const float* base = x + (long)blockIdx.z * L * ld;
  const int l0 = blockIdx.y * CM_ROWS, l1 = min(L, l0 + CM_ROWS);
  float s = 0.f;
  if (col < D)
    for (int l = l0 + rl; l < l1; l += 4) s += base[(long)l * ld + col];
  part[rl][threadIdx.x & 63] = s;
  __syncthreads();
  if (rl == 0 && col < D)
    partial[((long)blockIdx.z * gridDim.y + blockIdx.y) * D + col] =
        (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
}
__global__ void __launch_bounds__(256)
colmean_apply_kernel(float* __restrict__ x, const float* __restrict__ partial, int L, int D, long ld) {
  const int col = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rl = threadIdx.x >> 6;
  if (col >= D) return;
  float s = 0.f;
  for (int c = 0; c < (int)gridDim.y; ++c) s += partial[((long)blockIdx.z * gridDim.y + c) * D + col];
  const float mean = s / (float)L;
  float* base = x + (long)blockIdx.z * L * ld;
  const int l0 = blockIdx.y * CM_ROWS, l1 = min(L, l0 + CM_ROWS);
  for (int l = l0 + rl; l < l1; l += 4) base[(long)l * ld + col] -= mean;
}

// the same in ONE launch for short groups (L <= 1 024 rows: the 400 positions of a support map, the 49 of a RoI): one block
// per (64-column slab, group) walks the chunks itself -- the SAME chunked, ordered summation as the two-pass pair, hence the
// same bits -- and then subtracts; saves a dependent launch where the chain is latency-bound.
__global__ void __launch_bounds__(256)
colmean_fused_kernel(float* __restrict__ x, int L, int D, long ld) {
  __shared__ float part[4][64];
  __shared__ float mean_s[64];
  const int tc = threadIdx.x & 63, col = blockIdx.x * 64 + tc, rl = threadIdx.x >> 6;
  float* base = x + (long)blockIdx.z * L * ld;
  float total = 0.f;
  for (int l0 = 0; l0 < L; l0 += CM_ROWS) {
    const int l1 = min(L, l0 + CM_ROWS);
    float s = 0.f;
    if (col < D)
      for (int l = l0 + rl; l < l1; l += 4) s += base[(long)l * ld + col];
    part[rl][tc] = s;
    __syncthreads();
    if (rl == 0) total += (part[0][tc] + part[1][tc]) + (part[2][tc] + part[3][tc]);
    __syncthreads();
  }
  if (rl == 0) mean_s[tc] = total / (float)L;
  __syncthreads();
  if (col >= D) return;
  const float mean = mean_s[tc];
  for (int l = rl; l < L; l += 4) base[(long)l * ld + col] -= mean;
}

// batched transpose in[g][R][C] -> out[g][C][ldo] (32x32 LDS tiles); zero_pad: the grid covers ldo (not R) output
// columns and columns R..ldo-1 are written as zeros (a K-padded GEMM operand needs no separate fill launch)
__global__ void __launch_bounds__(256)
transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int R, int C, long ldi, long ldo,
                 long in_batch, long out_batch, int zero_pad) {
  __shared__ float t[32][33];
  const float* ib = in + (long)blockIdx.z * in_batch;
  float* ob = out + (long)blockIdx.z * out_batch;
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int j = ty; j < 32; j += 8) {
    const int r = r0 + j, c = c0 + tx;
    t[j][tx] = (r < R && c < C) ? ib[(long)r * ldi + c] : 0.f;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const int c = c0 + j, r = r0 + tx;
    if (c < C && (r < R || (zero_pad && r < ldo))) ob[(long)c * ldo + r] = t[tx][j];  // (t holds zeros for r >= R)
  }
}

// rois_label of the training forward (dana.py:191-194): labels of the positive-support head, zeros for the
// negative-support head, as int64 [2n] -- one launch instead of a float->long copy, a fill and a cat
__global__ void __launch_bounds__(256)
labels_posneg_kernel(const float* __restrict__ labels, long long* __restrict__ out, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 2 * n) return;
  out[i] = i < n ? (long long)labels[i] : 0ll;
}

// Adjoint of a TRAIN-mode BatchNorm (batch statistics; fgn.py:147-153's bn1 / bn2): with xhat = (x - mean) * istd,
//   dgamma = sum dy * xhat,  dbeta = sum dy,  dx = gamma * istd * (dy - dbeta / R - xhat * dgamma / R).
// One workgroup per 64 channels; 4 row lanes walk the rows, their partial sums are added in lane order (deterministic).
__global__ void __launch_bounds__(256)
bn_train_backward_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ mean,
                         const float* __restrict__ var, const float* __restrict__ gamma, float eps, long rows, int C,
                         float* __restrict__ dx, float* __restrict__ dgamma, float* __restrict__ dbeta, int accumulate) {
  __shared__ float s1[4][64], s2[4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const bool ok = c < C;
  const float mu = ok ? mean[c] : 0.f;
  const float istd = ok ? 1.0f / sqrtf(var[c] + eps) : 0.f;
  float a1 = 0.f, a2 = 0.f;
  if (ok)
    for (long r = rl; r < rows; r += 4) {
      const float g = dy[r * C + c];
      a1 += g;
      a2 += g * ((x[r * C + c] - mu) * istd);
    }
  s1[rl][cl] = a1;
  s2[rl][cl] = a2;
  __syncthreads();
  const float t1 = ((s1[0][cl] + s1[1][cl]) + s1[2][cl]) + s1[3][cl];
  const float t2 = ((s2[0][cl] + s2[1][cl]) + s2[2][cl]) + s2[3][cl];
  if (!ok) return;
  if (rl == 0) {
    dgamma[c] = accumulate ? dgamma[c] + t2 : t2;
    dbeta[c] = accumulate ? dbeta[c] + t1 : t1;
  }
  const float k = gamma[c] * istd, m1 = t1 / (float)rows, m2 = t2 / (float)rows;
  for (long r = rl; r < rows; r += 4) {
    const float xh = (x[r * C + c] - mu) * istd;
    dx[r * C + c] = k * (dy[r * C + c] - m1 - xh * m2);
  }
}

// out[m][n] = epi(alpha * sum_s partial[s][m][n]): the fixed-order reduction behind a split-K contraction
__global__ void __launch_bounds__(256)
splitk_reduce_kernel(const float* __restrict__ part, int S, long slab, int M, int N, float* __restrict__ out, long ldc,
                     const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ residual,
                     long ldr, float alpha, int relu) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;  // one float4 of the [M][N] result (N % 4 == 0)
  const int n4 = N >> 2;
  if (i >= (long)M * n4) return;
  const int m = (int)(i / n4), n = (int)(i - (long)m * n4) * 4;
  float4 acc = *(const float4*)(part + (long)m * N + n);
  for (int s = 1; s < S; ++s) {
    const float4 v = *(const float4*)(part + s * slab + (long)m * N + n);
    acc.x += v.x;
    acc.y += v.y;
    acc.z += v.z;
    acc.w += v.w;
  }
  float r[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    r[q] = r[q] * alpha * (scale ? scale[n + q] : 1.f) + (shift ? shift[n + q] : 0.f);
    if (residual) r[q] += residual[(long)m * ldr + n + q];
    if (relu) r[q] = fmaxf(r[q], 0.f);
  }
  float* o = out + (long)m * ldc + n;
  if ((ldc & 3) == 0 && (((uintptr_t)out) & 15) == 0) {
    *(float4*)o = make_float4(r[0], r[1], r[2], r[3]);
  } else {
    o[0] = r[0];
    o[1] = r[1];
    o[2] = r[2];
    o[3] = r[3];
  }
}

// [cout][k0 + k1] weight of a two-segment contraction with both frozen-BN scales folded into the rows
__global__ void pack_cat2_kernel(const float* __restrict__ w0, const float* __restrict__ s0, const float* __restrict__ b0,
                                 int k0, const float* __restrict__ w1, const float* __restrict__ s1,
                                 const float* __restrict__ b1, int k1, int cout, float* __restrict__ out,
                                 float* __restrict__ shift, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int kk = k0 + k1;
  const int n = (int)(i / kk), k = (int)(i - (long)n * kk);
  out[i] = k < k0 ? s0[n] * w0[(long)n * k0 + k] : s1[n] * w1[(long)n * k1 + (k - k0)];
  if (k == 0) shift[n] = b0[n] + b1[n];
}

// OIHW -> [O][KH][KW][I]
__global__ void __launch_bounds__(256)
pack_weight_kernel(const float* __restrict__ w, float* __restrict__ out, int I, int KH, int KW, long total) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)blockDim.x * gridDim.x) {
    const int ci = (int)(i % I);
    const int kw = (int)((i / I) % KW);
    const int kh = (int)((i / I / KW) % KH);
    const long o = i / I / KW / KH;
    out[i] = w[((o * I + ci) * KH + kh) * KW + kw];
  }
}

// stem: [O][3][7][7] -> [O][7][8][4] with zero taps/channels
__global__ void __launch_bounds__(256)
pack_stem_kernel(const float* __restrict__ w, float* __restrict__ out, int CI, long total) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)blockDim.x * gridDim.x) {
    const int ci = (int)(i % 4);
    const int kw = (int)((i / 4) % 8);
    const int kh = (int)((i / 32) % 7);
    const long o = i / 224;
    out[i] = (ci < CI && kw < 7) ? w[((o * CI + ci) * 7 + kh) * 7 + kw] : 0.f;
  }
}

__global__ void __launch_bounds__(256)
bn_fold_kernel(const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mean,
               const float* __restrict__ var, float eps, float* __restrict__ scale, float* __restrict__ shift,
               int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float s = gamma[i] / sqrtf(var[i] + eps);
  scale[i] = s;
  shift[i] = beta[i] - mean[i] * s;
}

}  // namespace

extern "C" {

int dana_nchw_to_nhwc(const float* in, float* out, int batch, int channels, int height, int width, int cpad,
                      long out_pix_stride, dana_stream_t stream) {
  DANA_CHECK_ARG(batch >= 0 && channels > 0 && height > 0 && width > 0 && cpad >= channels,
                 "dana_nchw_to_nhwc: bad shape");
  if (batch == 0) return DANA_OK;
  DANA_CHECK_ARG(in && out, "dana_nchw_to_nhwc: null pointer");
  if (out_pix_stride <= 0) out_pix_stride = cpad;
  const long total = (long)batch * height * width * cpad;
  if (channels == 3 && cpad == 4 && out_pix_stride == 4 && total / 4 < (1L << 31) && ((uintptr_t)out & 15) == 0) {
    const int pix = (int)(total / 4);
    nchw3_to_nhwc4_kernel<<<grid_for(pix, 256), 256, 0, (hipStream_t)stream>>>(in, (float4*)out, height * width, pix);
    DANA_CHECK_LAUNCH("dana_nchw_to_nhwc");
    return DANA_OK;
  }
  nchw_to_nhwc_kernel<<<grid_for(total, 256), 256, 0, (hipStream_t)stream>>>(in, out, channels, height * width, cpad,
                                                                             out_pix_stride, total);
  DANA_CHECK_LAUNCH("dana_nchw_to_nhwc");
  return DANA_OK;
}

int dana_nhwc_to_nchw(const float* in, float* out, int batch, int channels, int height, int width,
                      long in_pix_stride, dana_stream_t stream) {
  DANA_CHECK_ARG(batch >= 0 && channels > 0 && height > 0 && width > 0, "dana_nhwc_to_nchw: bad shape");
  if (batch == 0) return DANA_OK;
  DANA_CHECK_ARG(in && out, "dana_nhwc_to_nchw: null pointer");
  if (in_pix_stride <= 0) in_pix_stride = channels;
  const long total = (long)batch * height * width * channels;
  nhwc_to_nchw_kernel<<<grid_for(total, 256), 256, 0, (hipStream_t)stream>>>(in, out, channels, height * width,
                                                                             in_pix_stride, total);
  DANA_CHECK_LAUNCH("dana_nhwc_to_nchw");
  return DANA_OK;
}

int dana_maxpool3x3s2_ceil_nhwc(const float* in, float* out, int batch, int height, int width, int channels,
                                dana_stream_t stream) {
  DANA_CHECK_ARG(batch >= 0 && height >= 3 && width >= 3 && channels > 0 && channels % 4 == 0,
                 "dana_maxpool3x3s2_ceil_nhwc: bad shape");
  if (batch == 0) return DANA_OK;
  DANA_CHECK_ARG(in && out, "dana_maxpool3x3s2_ceil_nhwc: null pointer");
  int oh = (height - 3 + 1) / 2 + 1, ow = (width - 3 + 1) / 2 + 1;  // ceil((H-3)/2)+1
  if ((oh - 1) * 2 >= height) --oh;                                  // window must start inside the input
  if ((ow - 1) * 2 >= width) --ow;
  const long total = (long)batch * oh * ow * (channels / 4);
  maxpool3x3s2_kernel<<<grid_for(total, 256), 256, 0, (hipStream_t)stream>>>((const float4*)in, (float4*)out, height,
                                                                             width, oh, ow, channels / 4, total);
  DANA_CHECK_LAUNCH("dana_maxpool3x3s2_ceil_nhwc");
  return DANA_OK;
}

int dana_depthwise_corr_nhwc(const float* feat, const float* kernels, float* out, long n_maps, int height, int width,
                             int channels, int kh, int kw, long maps_per_kernel, long feat_pix_stride,
                             dana_stream_t stream) {
  DANA_CHECK_ARG(n_maps >= 0 && height >= kh && width >= kw && kh > 0 && kw > 0 && channels > 0 && channels % 4 == 0 &&
                     maps_per_kernel > 0,
                 "dana_depthwise_corr_nhwc: bad shape");
  if (n_maps == 0) return DANA_OK;
  DANA_CHECK_ARG(feat && kernels && out, "dana_depthwise_corr_nhwc: null pointer");
  if (feat_pix_stride <= 0) feat_pix_stride = channels;
  DANA_CHECK_ARG(feat_pix_stride % 4 == 0, "dana_depthwise_corr_nhwc: stride %% 4 != 0");
  const int oh = height - kh + 1, ow = width - kw + 1;
  const long total = n_maps * oh * ow * (channels / 4);
  depthwise_corr_kernel<<<grid_for(total, 256), 256, 0, (hipStream_t)stream>>>(
      (const float4*)feat, (const float4*)kernels, (float4*)out, height, width, kh, kw, oh, ow, channels / 4,
      feat_pix_stride / 4, maps_per_kernel, total);
  DANA_CHECK_LAUNCH("dana_depthwise_corr_nhwc");
  return DANA_OK;
}

int dana_depthwise_corr_backward_nhwc(const float* grad_out, const float* feat, const float* kernels, float* grad_feat,
                                      float* grad_kernels, long n_maps, int height, int width, int channels, int kh,
                                      int kw, long maps_per_kernel, long feat_pix_stride, int accumulate_kernels,
                                      dana_stream_t stream) {
  DANA_CHECK_ARG(n_maps >= 0 && height >= kh && width >= kw && kh > 0 && kw > 0 && channels > 0 && channels % 4 == 0 &&
                     maps_per_kernel > 0,
                 "dana_depthwise_corr_backward_nhwc: bad shape");
  if (n_maps == 0) return DANA_OK;
  DANA_CHECK_ARG(grad_out && (!grad_feat || kernels) && (!grad_kernels || feat),
                 "dana_depthwise_corr_backward_nhwc: null pointer");
  if (feat_pix_stride <= 0) feat_pix_stride = channels;
  DANA_CHECK_ARG(feat_pix_stride % 4 == 0, "dana_depthwise_corr_backward_nhwc: stride %% 4 != 0");
  const int oh = height - kh + 1, ow = width - kw + 1, c4 = channels / 4;
  hipStream_t s = (hipStream_t)stream;
  if (grad_feat) {
    const long total = n_maps * height * width * c4;
    depthwise_corr_bwd_feat_kernel<<<grid_for(total, 256), 256, 0, s>>>((const float4*)grad_out, (const float4*)kernels,
                                                                        (float4*)grad_feat, height, width, kh, kw, oh, ow,
                                                                        c4, maps_per_kernel, total);
  }
  if (grad_kernels) {
    const long nk = (n_maps + maps_per_kernel - 1) / maps_per_kernel;
    const long total = nk * kh * kw * c4;
    depthwise_corr_bwd_kern_kernel<<<grid_for(total, 256), 256, 0, s>>>(
        (const float4*)grad_out, (const float4*)feat, (float4*)grad_kernels, height, width, kh, kw, oh, ow, c4,
        feat_pix_stride / 4, maps_per_kernel, n_maps, total, accumulate_kernels);
  }
  DANA_CHECK_LAUNCH("dana_depthwise_corr_backward_nhwc");
  return DANA_OK;
}

int dana_maxpool2x2s2_nhwc(const float* in, float* out, int batch, int height, int width, int channels,
                           dana_stream_t stream) {
  DANA_CHECK_ARG(batch >= 0 && height >= 2 && width >= 2 && channels > 0 && channels % 4 == 0,
                 "dana_maxpool2x2s2_nhwc: bad shape");
  if (batch == 0) return DANA_OK;
  DANA_CHECK_ARG(in && out, "dana_maxpool2x2s2_nhwc: null pointer");
  const int oh = height / 2, ow = width / 2;
  const long total = (long)batch * oh * ow * (channels / 4);
  maxpool2x2s2_kernel<<<grid_for(total, 256), 256, 0, (hipStream_t)stream>>>((const float4*)in, (float4*)out, height,
                                                                             width, oh, ow, channels / 4, total);
  DANA_CHECK_LAUNCH("dana_maxpool2x2s2_nhwc");
  return DANA_OK;
}

int dana_maxpool2x2s2_backward_nhwc(const float* in, const float* grad_out, float* grad_in, int batch, int height,
                                    int width, int channels, dana_stream_t stream) {
  DANA_CHECK_ARG(batch >= 0 && height >= 2 && width >= 2 && channels > 0, "dana_maxpool2x2s2_backward_nhwc: bad shape");
  if (batch == 0) return DANA_OK;
  DANA_CHECK_ARG(in && grad_out && grad_in, "dana_maxpool2x2s2_backward_nhwc: null pointer");
  const long total = (long)batch * height * width * channels;
  maxpool2x2s2_bwd_kernel<<<grid_for(total, 256), 256, 0, (hipStream_t)stream>>>(in, grad_out, grad_in, height, width,
                                                                                 height / 2, width / 2, channels, total);
  DANA_CHECK_LAUNCH("dana_maxpool2x2s2_backward_nhwc");
  return DANA_OK;
}

int dana_sigmoid(float* x, long n, dana_stream_t stream) {
  DANA_CHECK_ARG(n >= 0, "dana_sigmoid: bad size");
  if (n == 0) return DANA_OK;
  DANA_CHECK_ARG(x, "dana_sigmoid: null pointer");
  sigmoid_kernel<<<grid_for(n, 256), 256, 0, (hipStream_t)stream>>>(x, n);
  DANA_CHECK_LAUNCH("dana_sigmoid");
  return DANA_OK;
}

int dana_scale_rows_by_group(const float* x, const float* group_vec, float* out, long rows, long rows_per_group,
                             int channels, dana_stream_t stream) {
  DANA_CHECK_ARG(rows >= 0 && rows_per_group > 0 && channels > 0 && channels % 4 == 0,
                 "dana_scale_rows_by_group: bad shape");
  if (rows == 0) return DANA_OK;
  DANA_CHECK_ARG(x && group_vec && out, "dana_scale_rows_by_group: null pointer");
  const long total = rows * (channels / 4);
  scale_rows_by_group_kernel<<<grid_for(total, 256), 256, 0, (hipStream_t)stream>>>(
      (const float4*)x, (const float4*)group_vec, (float4*)out, rows_per_group, channels / 4, total);
  DANA_CHECK_LAUNCH("dana_scale_rows_by_group");
  return DANA_OK;
}

int dana_avgpool_nhwc(const float* in, float* out, int batch, int height, int width, int channels, int k, int stride,
                      dana_stream_t stream) {
  DANA_CHECK_ARG(batch >= 0 && k > 0 && stride > 0 && height >= k && width >= k && channels % 4 == 0,
                 "dana_avgpool_nhwc: bad shape");
  if (batch == 0) return DANA_OK;
  DANA_CHECK_ARG(in && out, "dana_avgpool_nhwc: null pointer");
  const int oh = (height - k) / stride + 1, ow = (width - k) / stride + 1;
  const long total = (long)batch * oh * ow * (channels / 4);
  const size_t tile_lds = (size_t)(height * width + height * ow) * AP_CH4 * sizeof(float4);
  if (stride == 1 && (channels / 4) % AP_CH4 == 0 && tile_lds <= 48 * 1024 && height * width >= 64) {
    avgpool_tile_kernel<<<(unsigned)((long)batch * (channels / 4 / AP_CH4)), 256, tile_lds, (hipStream_t)stream>>>(
        (const float4*)in, (float4*)out, height, width, oh, ow, channels / 4, k);
  } else if (stride == 1 && ow <= 8) {
    const long rows_total = (long)batch * oh * (channels / 4);
    avgpool_rows_kernel<<<grid_for(rows_total, 256), 256, 0, (hipStream_t)stream>>>(
        (const float4*)in, (float4*)out, height, width, oh, ow, channels / 4, k, rows_total);
  } else {
    avgpool_kernel<<<grid_for(total, 256), 256, 0, (hipStream_t)stream>>>((const float4*)in, (float4*)out, height, width,
                                                                          oh, ow, channels / 4, k, stride, total);
  }
  DANA_CHECK_LAUNCH("dana_avgpool_nhwc");
  return DANA_OK;
}

int dana_spatial_mean_nhwc(const float* in, float* out, int groups, int positions, int channels, long in_pix_stride,
                           dana_stream_t stream) {
  DANA_CHECK_ARG(groups >= 0 && positions > 0 && channels % 4 == 0, "dana_spatial_mean_nhwc: bad shape");
  if (groups == 0) return DANA_OK;
  DANA_CHECK_ARG(in && out, "dana_spatial_mean_nhwc: null pointer");
  if (in_pix_stride <= 0) in_pix_stride = channels;
  DANA_CHECK_ARG(in_pix_stride % 4 == 0, "dana_spatial_mean_nhwc: stride %% 4 != 0");
  const long total = (long)groups * (channels / 4);
  spatial_mean_kernel<<<grid_for(total, 256), 256, 0, (hipStream_t)stream>>>(
      (const float4*)in, (float4*)out, positions, channels / 4, in_pix_stride / 4, total);
  DANA_CHECK_LAUNCH("dana_spatial_mean_nhwc");
  return DANA_OK;
}

int dana_add_pe(const float* in, const float* pe, float* out, long rows, int length, int channels,
                long in_stride, long out_stride, dana_stream_t stream) {
  DANA_CHECK_ARG(rows >= 0 && length > 0 && channels > 0 && channels % 4 == 0, "dana_add_pe: bad shape");
  if (rows == 0) return DANA_OK;
  DANA_CHECK_ARG(in && pe && out, "dana_add_pe: null pointer");
  if (in_stride <= 0) in_stride = channels;
  if (out_stride <= 0) out_stride = channels;
  DANA_CHECK_ARG(in_stride % 4 == 0 && out_stride % 4 == 0, "dana_add_pe: strides %% 4 != 0");
  const long total = rows * (channels / 4);
  add_pe_kernel<<<grid_for(total, 256), 256, 0, (hipStream_t)stream>>>(
      (const float4*)in, (const float4*)pe, (float4*)out, length, channels / 4, in_stride / 4, out_stride / 4, total);
  DANA_CHECK_LAUNCH("dana_add_pe");
  return DANA_OK;
}

int dana_add_pe_groups(const float* in, const float* pe, float* out, long groups, long rows_per_group, int length, int channels,
                       long in_group_stride, long out_group_stride, dana_stream_t stream) {
  DANA_CHECK_ARG(groups >= 0 && rows_per_group > 0 && length > 0 && channels > 0 && channels % 4 == 0 &&
                     in_group_stride % 4 == 0 && out_group_stride % 4 == 0,
                 "dana_add_pe_groups: bad shape");
  if (groups == 0) return DANA_OK;
  DANA_CHECK_ARG(in && pe && out, "dana_add_pe_groups: null pointer");
  const long total = groups * rows_per_group * (channels / 4);
  add_pe_groups_kernel<<<grid_for(total, 256), 256, 0, (hipStream_t)stream>>>(
      (const float4*)in, (const float4*)pe, (float4*)out, length, channels / 4, rows_per_group, in_group_stride / 4,
      out_group_stride / 4, total);
  DANA_CHECK_LAUNCH("dana_add_pe_groups");
  return DANA_OK;
}

size_t dana_colmean_sub_workspace_bytes(int groups, int length, int dim) {
  if (groups <= 0 || length <= 0 || dim <= 0) return 0;
  return (size_t)groups * ((length + CM_ROWS - 1) / CM_ROWS) * dim * sizeof(float);
}

int dana_colmean_sub(float* x, int groups, int length, int dim, long ld, void* workspace, size_t workspace_bytes,
                     dana_stream_t stream) {
  DANA_CHECK_ARG(groups >= 0 && length > 0 && dim > 0, "dana_colmean_sub: bad shape");
  if (groups == 0) return DANA_OK;
  DANA_CHECK_ARG(x, "dana_colmean_sub: null pointer");
  if (ld <= 0) ld = dim;
  const size_t need = dana_colmean_sub_workspace_bytes(groups, length, dim);
  if (!workspace || workspace_bytes < need) {
    dana_set_error("dana_colmean_sub: workspace %zu < %zu", workspace_bytes, need);
    return DANA_ERR_WORKSPACE;
  }
  if (length <= 1024 && groups <= 65535) {  // short groups: one launch, same summation order (colmean_fused_kernel)
    dim3 grid1(dana_ceil_div(dim, 64), 1, groups);
    colmean_fused_kernel<<<grid1, 256, 0, (hipStream_t)stream>>>(x, length, dim, ld);
    DANA_CHECK_LAUNCH("dana_colmean_sub(fused)");
    return DANA_OK;
  }
  dim3 grid(dana_ceil_div(dim, 64), dana_ceil_div(length, CM_ROWS), groups);
  DANA_CHECK_ARG(grid.y <= 65535 && grid.z <= 65535, "dana_colmean_sub: too many chunks/groups");
  colmean_partial_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(x, (float*)workspace, length, dim, ld);
  DANA_CHECK_LAUNCH("dana_colmean_sub(partial)");
  colmean_apply_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(x, (const float*)workspace, length, dim, ld);
  DANA_CHECK_LAUNCH("dana_colmean_sub(apply)");
  return DANA_OK;
}

int dana_transpose_batched(const float* in, float* out, int groups, int rows, int cols, long ldi, long ldo,
                           long in_batch, long out_batch, dana_stream_t stream) {
  DANA_CHECK_ARG(groups >= 0 && rows > 0 && cols > 0 && ldi >= cols && ldo >= rows, "dana_transpose_batched: bad shape");
  if (groups == 0) return DANA_OK;
  DANA_CHECK_ARG(in && out, "dana_transpose_batched: null pointer");
  const int zero_pad = ldo > rows ? 1 : 0;  // (the output rows' tail is part of the result: zeros)
  dim3 grid(dana_ceil_div(cols, 32), dana_ceil_div(zero_pad ? (int)ldo : rows, 32), groups);
  transpose_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(in, out, rows, cols, ldi, ldo, in_batch, out_batch, zero_pad);
  DANA_CHECK_LAUNCH("dana_transpose_batched");
  return DANA_OK;
}

int dana_labels_posneg_i64(const float* labels, long long* out, long n, dana_stream_t stream) {
  DANA_CHECK_ARG(n >= 0, "dana_labels_posneg_i64: bad shape");
  if (n == 0) return DANA_OK;
  DANA_CHECK_ARG(labels && out, "dana_labels_posneg_i64: null pointer");
  labels_posneg_kernel<<<dana_ceil_div(2 * n, 256), 256, 0, (hipStream_t)stream>>>(labels, out, n);
  DANA_CHECK_LAUNCH("dana_labels_posneg_i64");
  return DANA_OK;
}

int dana_pack_conv_weight(const float* w_oihw, float* out, int cout, int cin, int kh, int kw, int stem7,
                          dana_stream_t stream) {
  DANA_CHECK_ARG(cout > 0 && cin > 0 && kh > 0 && kw > 0, "dana_pack_conv_weight: bad shape");
  DANA_CHECK_ARG(w_oihw && out, "dana_pack_conv_weight: null pointer");
  if (stem7) {
    DANA_CHECK_ARG(kh == 7 && kw == 7 && cin <= 4, "dana_pack_conv_weight: stem7 needs 7x7, cin<=4");
    const long total = (long)cout * 224;
    pack_stem_kernel<<<grid_for(total, 256), 256, 0, (hipStream_t)stream>>>(w_oihw, out, cin, total);
  } else {
    const long total = (long)cout * cin * kh * kw;
    pack_weight_kernel<<<grid_for(total, 256), 256, 0, (hipStream_t)stream>>>(w_oihw, out, cin, kh, kw, total);
  }
  DANA_CHECK_LAUNCH("dana_pack_conv_weight");
  return DANA_OK;
}

int dana_bn_train_backward(const float* grad_out, const float* x, const float* mean, const float* var_biased,
                           const float* gamma, float eps, long rows, int channels, float* grad_x, float* grad_gamma,
                           float* grad_beta, int accumulate, dana_stream_t stream) {
  DANA_CHECK_ARG(rows > 0 && channels > 0, "dana_bn_train_backward: bad shape");
  DANA_CHECK_ARG(grad_out && x && mean && var_biased && gamma && grad_x && grad_gamma && grad_beta,
                 "dana_bn_train_backward: null pointer");
  bn_train_backward_kernel<<<dana_ceil_div(channels, 64), 256, 0, (hipStream_t)stream>>>(
      grad_out, x, mean, var_biased, gamma, eps, rows, channels, grad_x, grad_gamma, grad_beta, accumulate);
  DANA_CHECK_LAUNCH("dana_bn_train_backward");
  return DANA_OK;
}

int dana_splitk_reduce(const float* partials, int slices, int m, int n, float* out, long ldc, const float* scale,
                       const float* shift, const float* residual, long ldr, float alpha, int flags,
                       dana_stream_t stream) {
  DANA_CHECK_ARG(slices > 0 && m >= 0 && n > 0 && n % 4 == 0 && ldc >= n, "dana_splitk_reduce: bad shape");
  if (m == 0) return DANA_OK;
  DANA_CHECK_ARG(partials && out && (((uintptr_t)partials) & 15) == 0, "dana_splitk_reduce: bad pointer");
  const long total = (long)m * (n / 4);
  splitk_reduce_kernel<<<grid_for(total, 256), 256, 0, (hipStream_t)stream>>>(
      partials, slices, (long)m * n, m, n, out, ldc, scale, shift, residual, ldr > 0 ? ldr : ldc, alpha,
      (flags & DANA_EPI_RELU) ? 1 : 0);
  DANA_CHECK_LAUNCH("dana_splitk_reduce");
  return DANA_OK;
}

int dana_pack_cat2_weight(const float* w0, const float* s0, const float* b0, int k0, const float* w1, const float* s1,
                          const float* b1, int k1, int cout, float* w_cat, float* shift, dana_stream_t stream) {
  DANA_CHECK_ARG(k0 > 0 && k1 > 0 && cout > 0 && w0 && w1 && s0 && s1 && b0 && b1 && w_cat && shift,
                 "dana_pack_cat2_weight: bad args");
  const long total = (long)cout * (k0 + k1);
  pack_cat2_kernel<<<grid_for(total, 256), 256, 0, (hipStream_t)stream>>>(w0, s0, b0, k0, w1, s1, b1, k1, cout, w_cat,
                                                                          shift, total);
  DANA_CHECK_LAUNCH("dana_pack_cat2_weight");
  return DANA_OK;
}

int dana_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var, float eps, float* scale,
                 float* shift, int n, dana_stream_t stream) {
  DANA_CHECK_ARG(n > 0 && gamma && beta && mean && var && scale && shift, "dana_bn_fold: bad args");
  bn_fold_kernel<<<dana_ceil_div(n, 256), 256, 0, (hipStream_t)stream>>>(gamma, beta, mean, var, eps, scale, shift, n);
  DANA_CHECK_LAUNCH("dana_bn_fold");
  return DANA_OK;
}

}  // extern "C"
