// Small element-wise / reduction kernels of the backward pass (training step groundwork, SURVEY.md 8d
// variant S). The contractions of the backward (data and weight gradients) run on igemm.hip / wgrad.hip;
// these are the HBM-bound glue between them: ReLU masks, frozen-BN row scaling of weight gradients, bias
// gradients (column sums), packed -> OIHW weight-gradient layout, pooling adjoints, softmax adjoints.
#include "common.h"
#include "../../include/dana_hip.h"

namespace {

int grid_for(long total, int block) {
  long g = (total + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > 65535L * 16 ? 65535L * 16 : g));
}

// g[r][c] = act[r][c] > 0 ? g[r][c] : 0   (adjoint of ReLU; act = the saved layer OUTPUT)
__global__ void __launch_bounds__(256)
relu_mask_kernel(float4* __restrict__ g, const float4* __restrict__ act, int C4, long ldg4, long lda4, long total) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)blockDim.x * gridDim.x) {
    const long r = i / C4;
    const int c = (int)(i % C4);
    float4 v = g[r * ldg4 + c];
    const float4 a = act[r * lda4 + c];
    v.x = a.x > 0.f ? v.x : 0.f;
    v.y = a.y > 0.f ? v.y : 0.f;
    v.z = a.z > 0.f ? v.z : 0.f;
    v.w = a.w > 0.f ? v.w : 0.f;
    g[r * ldg4 + c] = v;
  }
}

// y[r][c] (+)= alpha * x[r][c]
__global__ void __launch_bounds__(256)
axpy_kernel(float4* __restrict__ y, const float4* __restrict__ x, int C4, long ldy4, long ldx4, long total, float alpha,
            int accumulate) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)blockDim.x * gridDim.x) {
    const long r = i / C4;
    const int c = (int)(i % C4);
    const float4 a = x[r * ldx4 + c];
    float4 v = accumulate ? y[r * ldy4 + c] : make_float4(0.f, 0.f, 0.f, 0.f);
    v.x += alpha * a.x;
    v.y += alpha * a.y;
    v.z += alpha * a.z;
    v.w += alpha * a.w;
    y[r * ldy4 + c] = v;
  }
}

// y[r][c] *= x[r][c] (attention_type 'product': dana.py:155-156, 285-286)
__global__ void __launch_bounds__(256)
mul_rows_kernel(float4* __restrict__ y, const float4* __restrict__ x, int C4, long ldy4, long ldx4, long total) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)blockDim.x * gridDim.x) {
    const long r = i / C4;
    const int c = (int)(i % C4);
    const float4 a = x[r * ldx4 + c];
    float4 v = y[r * ldy4 + c];
    v.x *= a.x;
    v.y *= a.y;
    v.z *= a.z;
    v.w *= a.w;
    y[r * ldy4 + c] = v;
  }
}

// dw[n][:] *= scale[n]
__global__ void __launch_bounds__(256)
rowscale_kernel(float* __restrict__ dw, const float* __restrict__ scale, long K, long total) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)blockDim.x * gridDim.x)
    dw[i] *= scale[i / K];
}

// packed [O][KH][KW][I] -> OIHW (+= into the parameter's .grad when accumulate)
__global__ void __launch_bounds__(256)
unpack_weight_kernel(const float* __restrict__ packed, float* __restrict__ w, int I, int KH, int KW, long total,
                     int accumulate) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)blockDim.x * gridDim.x) {
    const int kw = (int)(i % KW);
    const int kh = (int)((i / KW) % KH);
    const int ci = (int)((i / KW / KH) % I);
    const long o = i / KW / KH / I;
    const float v = packed[((o * KH + kh) * KW + kw) * I + ci];
    w[i] = accumulate ? w[i] + v : v;
  }
}

// out[c] (+)= sum_r x[r][c]  -- two-stage deterministic: grid (C/64, chunks), then a tiny finisher
constexpr int CS_ROWS = 256;
__global__ void __launch_bounds__(256)
colsum_partial_kernel(const float* __restrict__ x, float* __restrict__ partial, long rows, int C, long ld, long x_batch = 0) {
  __shared__ float part[4][64];
  x += (long)blockIdx.z * x_batch;  // (batched: one matrix per blockIdx.z, its partial sums behind the previous one's)
  partial += (long)blockIdx.z * gridDim.y * C;
  const int col = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rl = threadIdx.x >> 6;
  const long r0 = (long)blockIdx.y * CS_ROWS, r1 = min(rows, r0 + CS_ROWS);
  float s = 0.f;
  if (col < C)
    for (long r = r0 + rl; r < r1; r += 4) s += x[r * ld + col];
  part[rl][threadIdx.x & 63] = s;
  __syncthreads();
  if (rl == 0 && col < C)
    partial[(long)blockIdx.y * C + col] = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
}
// partial[chunk][c] = sum over the chunk's rows of (x[r][c] - mean[c])^2  (BatchNorm batch variance, second pass)
__global__ void __launch_bounds__(256)
colsqdev_partial_kernel(const float* __restrict__ x, const float* __restrict__ mean, float* __restrict__ partial,
                        long rows, int C, long ld) {
  __shared__ float part[4][64];
  const int col = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rl = threadIdx.x >> 6;
  const long r0 = (long)blockIdx.y * CS_ROWS, r1 = min(rows, r0 + CS_ROWS);
  float s = 0.f;
  if (col < C) {
    const float m = mean[col];
    for (long r = r0 + rl; r < r1; r += 4) {
      const float d = x[r * ld + col] - m;
      s += d * d;
    }
  }
  part[rl][threadIdx.x & 63] = s;
  __syncthreads();
  if (rl == 0 && col < C)
    partial[(long)blockIdx.y * C + col] = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
}

// x[r][c] = relu?(x[r][c] * scale[c] + shift[c]), in place (a BatchNorm applied after the conv's residual sum)
__global__ void __launch_bounds__(256)
scale_shift_relu_kernel(float4* __restrict__ x, const float4* __restrict__ scale, const float4* __restrict__ shift,
                        long total, int C4, int relu) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)blockDim.x * gridDim.x) {
    const int c = (int)(i % C4);
    const float4 v = x[i], sc = scale[c], sh = shift[c];
    float4 o = make_float4(v.x * sc.x + sh.x, v.y * sc.y + sh.y, v.z * sc.z + sh.z, v.w * sc.w + sh.w);
    if (relu) o = make_float4(fmaxf(o.x, 0.f), fmaxf(o.y, 0.f), fmaxf(o.z, 0.f), fmaxf(o.w, 0.f));
    x[i] = o;
  }
}

__global__ void __launch_bounds__(256)
colsum_final_kernel(const float* __restrict__ partial, float* __restrict__ out, int chunks, int C, float alpha,
                    int accumulate, long out_batch = 0) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  partial += (long)blockIdx.y * chunks * C;
  out += (long)blockIdx.y * out_batch;
  float s = 0.f;
  for (int k = 0; k < chunks; ++k) s += partial[(long)k * C + c];
  out[c] = (accumulate ? out[c] : 0.f) + alpha * s;
}

// adjoint of the k x k / stride average pool: gin[b][h][w][c] = (1/k^2) * sum of gout over the windows covering (h, w)
__global__ void __launch_bounds__(256)
avgpool_bwd_kernel(const float4* __restrict__ gout, float4* __restrict__ gin, int H, int W, int OH, int OW, int C4, int k,
                   int stride, long total) {
  const float inv = 1.f / (float)(k * k);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)blockDim.x * gridDim.x) {
    const int c = (int)(i % C4);
    const int w = (int)((i / C4) % W);
    const int h = (int)((i / C4 / W) % H);
    const long b = i / C4 / W / H;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int oh = 0; oh < OH; ++oh) {
      if (h < oh * stride || h >= oh * stride + k) continue;
      for (int ow = 0; ow < OW; ++ow) {
        if (w < ow * stride || w >= ow * stride + k) continue;
        const float4 v = gout[((b * OH + oh) * OW + ow) * C4 + c];
        s.x += v.x;
        s.y += v.y;
        s.z += v.z;
        s.w += v.w;
      }
    }
    gin[i] = make_float4(s.x * inv, s.y * inv, s.z * inv, s.w * inv);
  }
}

// adjoint of softmax over the last dim, in place on g: g <- p * (g - sum(p * g)); one wave per row
__global__ void __launch_bounds__(256)
softmax_bwd_rows_kernel(float* __restrict__ g, const float* __restrict__ p, long rows, int L, long ldg, long ldp) {
  const int lane = threadIdx.x & 63;
  const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  float* gr = g + r * ldg;
  const float* pr = p + r * ldp;
  float dot = 0.f;
  for (int l = lane; l < L; l += 64) dot += gr[l] * pr[l];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) dot += __shfl_xor(dot, o);
  for (int l = lane; l < L; l += 64) gr[l] = pr[l] * (gr[l] - dot);
}

// tiny general GEMM for the skinny heads (N or K of 2 / 4 / 72): c[m][n] (+)= alpha * sum_k a(m,k) * b(k,n) with
// arbitrary element strides; one lane per output element -- only for problems of a few MFLOP
__global__ void __launch_bounds__(256)
gemm_small_kernel(const float* __restrict__ a, long sam, long sak, const float* __restrict__ b, long sbk, long sbn,
                  float* __restrict__ c, long scm, long scn, int M, int N, int K, float alpha, int accumulate) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)M * N) return;
  const int n = (int)(i % N);
  const long m = i / N;
  float s = 0.f;
  for (int k = 0; k < K; ++k) s += a[m * sam + k * sak] * b[k * sbk + n * sbn];
  float* o = c + m * scm + n * scn;
  *o = (accumulate ? *o : 0.f) + alpha * s;
}

// long-K variant (K >= 64, e.g. the [2 x n_roi] x [n_roi x 1024] weight gradients of the skinny heads): a block =
// 64 output columns of one output row x 4 K-slices; the slices are summed through LDS in a fixed order
__global__ void __launch_bounds__(256)
gemm_small_ksplit_kernel(const float* __restrict__ a, long sam, long sak, const float* __restrict__ b, long sbk, long sbn,
                         float* __restrict__ c, long scm, long scn, int N, int K, float alpha, int accumulate) {
  __shared__ float part[4][64];
  const int nl = threadIdx.x & 63, ks = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + nl;
  const long m = blockIdx.y;
  float s0 = 0.f, s1 = 0.f;
  if (n < N) {
    int k = ks;
    for (; k + 4 < K; k += 8) {  // two independent chains per lane
      s0 += a[m * sam + k * sak] * b[k * sbk + n * sbn];
      s1 += a[m * sam + (k + 4) * sak] * b[(k + 4) * sbk + n * sbn];
    }
    for (; k < K; k += 4) s0 += a[m * sam + k * sak] * b[k * sbk + n * sbn];
  }
  part[ks][nl] = s0 + s1;
  __syncthreads();
  if (ks == 0 && n < N) {
    const float s = (part[0][nl] + part[1][nl]) + (part[2][nl] + part[3][nl]);
    float* o = c + m * scm + n * scn;
    *o = (accumulate ? *o : 0.f) + alpha * s;
  }
}

}  // namespace

extern "C" {

int dana_relu_mask(float* grad, const float* act, long rows, int channels, long ld_grad, long ld_act,
                   dana_stream_t stream) {
  DANA_CHECK_ARG(rows >= 0 && channels > 0 && channels % 4 == 0, "dana_relu_mask: bad shape");
  if (rows == 0) return DANA_OK;
  DANA_CHECK_ARG(grad && act, "dana_relu_mask: null pointer");
  if (ld_grad <= 0) ld_grad = channels;
  if (ld_act <= 0) ld_act = channels;
  DANA_CHECK_ARG(ld_grad % 4 == 0 && ld_act % 4 == 0, "dana_relu_mask: strides %% 4 != 0");
  const long total = rows * (channels / 4);
  relu_mask_kernel<<<grid_for(total, 256), 256, 0, (hipStream_t)stream>>>((float4*)grad, (const float4*)act, channels / 4,
                                                                          ld_grad / 4, ld_act / 4, total);
  DANA_CHECK_LAUNCH("dana_relu_mask");
  return DANA_OK;
}

int dana_axpy_rows(float* y, const float* x, long rows, int channels, long ld_y, long ld_x, float alpha,
                   int accumulate, dana_stream_t stream) {
  DANA_CHECK_ARG(rows >= 0 && channels > 0 && channels % 4 == 0, "dana_axpy_rows: bad shape");
  if (rows == 0) return DANA_OK;
  DANA_CHECK_ARG(y && x, "dana_axpy_rows: null pointer");
  if (ld_y <= 0) ld_y = channels;
  if (ld_x <= 0) ld_x = channels;
  DANA_CHECK_ARG(ld_y % 4 == 0 && ld_x % 4 == 0, "dana_axpy_rows: strides %% 4 != 0");
  const long total = rows * (channels / 4);
  axpy_kernel<<<grid_for(total, 256), 256, 0, (hipStream_t)stream>>>((float4*)y, (const float4*)x, channels / 4, ld_y / 4,
                                                                     ld_x / 4, total, alpha, accumulate);
  DANA_CHECK_LAUNCH("dana_axpy_rows");
  return DANA_OK;
}

int dana_mul_rows(float* y, const float* x, long rows, int channels, long ld_y, long ld_x, dana_stream_t stream) {
  DANA_CHECK_ARG(y && x && rows > 0 && channels > 0 && channels % 4 == 0, "dana_mul_rows: bad args");
  if (ld_y <= 0) ld_y = channels;
  if (ld_x <= 0) ld_x = channels;
  DANA_CHECK_ARG(ld_y % 4 == 0 && ld_x % 4 == 0 && (((uintptr_t)y | (uintptr_t)x) & 15) == 0, "dana_mul_rows: strides %% 4 != 0 or unaligned rows");
  const long total = rows * (channels / 4);
  mul_rows_kernel<<<grid_for(total, 256), 256, 0, (hipStream_t)stream>>>((float4*)y, (const float4*)x, channels / 4, ld_y / 4,
                                                                         ld_x / 4, total);
  DANA_CHECK_LAUNCH("dana_mul_rows");
  return DANA_OK;
}

int dana_rowscale(float* dw, const float* scale, int rows, long cols, dana_stream_t stream) {
  DANA_CHECK_ARG(dw && scale && rows > 0 && cols > 0, "dana_rowscale: bad args");
  const long total = (long)rows * cols;
  rowscale_kernel<<<grid_for(total, 256), 256, 0, (hipStream_t)stream>>>(dw, scale, cols, total);
  DANA_CHECK_LAUNCH("dana_rowscale");
  return DANA_OK;
}

int dana_unpack_conv_weight_grad(const float* packed, float* w_oihw, int cout, int cin, int kh, int kw, int accumulate,
                                 dana_stream_t stream) {
  DANA_CHECK_ARG(packed && w_oihw && cout > 0 && cin > 0 && kh > 0 && kw > 0, "dana_unpack_conv_weight_grad: bad args");
  const long total = (long)cout * cin * kh * kw;
  unpack_weight_kernel<<<grid_for(total, 256), 256, 0, (hipStream_t)stream>>>(packed, w_oihw, cin, kh, kw, total,
                                                                              accumulate);
  DANA_CHECK_LAUNCH("dana_unpack_conv_weight_grad");
  return DANA_OK;
}

size_t dana_colsum_workspace_bytes(long rows, int channels) {
  if (rows <= 0 || channels <= 0) return 0;
  return (size_t)((rows + CS_ROWS - 1) / CS_ROWS) * channels * sizeof(float);
}

int dana_colsum(const float* x, float* out, long rows, int channels, long ld, float alpha, int accumulate,
                void* workspace, size_t workspace_bytes, dana_stream_t stream) {
  DANA_CHECK_ARG(rows > 0 && channels > 0 && x && out, "dana_colsum: bad args");
  if (ld <= 0) ld = channels;
  const size_t need = dana_colsum_workspace_bytes(rows, channels);
  if (!workspace || workspace_bytes < need) {
    dana_set_error("dana_colsum: workspace %zu < %zu", workspace_bytes, need);
    return DANA_ERR_WORKSPACE;
  }
  const int chunks = (int)((rows + CS_ROWS - 1) / CS_ROWS);
  DANA_CHECK_ARG(chunks <= 65535, "dana_colsum: too many rows");
  dim3 grid(dana_ceil_div(channels, 64), chunks);
  colsum_partial_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(x, (float*)workspace, rows, channels, ld);
  DANA_CHECK_LAUNCH("dana_colsum(partial)");
  colsum_final_kernel<<<dana_ceil_div(channels, 256), 256, 0, (hipStream_t)stream>>>((const float*)workspace, out, chunks,
                                                                                      channels, alpha, accumulate);
  DANA_CHECK_LAUNCH("dana_colsum(final)");
  return DANA_OK;
}

int dana_colsum_batched(const float* x, float* out, int batch, long rows, int channels, long ld, long x_batch, long out_batch,
                        float alpha, int accumulate, void* workspace, size_t workspace_bytes, dana_stream_t stream) {
  DANA_CHECK_ARG(batch >= 0 && rows > 0 && channels > 0, "dana_colsum_batched: bad args");
  if (batch == 0) return DANA_OK;
  DANA_CHECK_ARG(x && out && batch <= 65535, "dana_colsum_batched: bad args");
  if (ld <= 0) ld = channels;
  const size_t need = (size_t)batch * dana_colsum_workspace_bytes(rows, channels);
  if (!workspace || workspace_bytes < need) {
    dana_set_error("dana_colsum_batched: workspace %zu < %zu", workspace_bytes, need);
    return DANA_ERR_WORKSPACE;
  }
  const int chunks = (int)((rows + CS_ROWS - 1) / CS_ROWS);
  DANA_CHECK_ARG(chunks <= 65535, "dana_colsum_batched: too many rows");
  dim3 grid(dana_ceil_div(channels, 64), chunks, batch);
  colsum_partial_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(x, (float*)workspace, rows, channels, ld, x_batch);
  DANA_CHECK_LAUNCH("dana_colsum_batched(partial)");
  dim3 fgrid(dana_ceil_div(channels, 256), batch);
  colsum_final_kernel<<<fgrid, 256, 0, (hipStream_t)stream>>>((const float*)workspace, out, chunks, channels, alpha,
                                                               accumulate, out_batch);
  DANA_CHECK_LAUNCH("dana_colsum_batched(final)");
  return DANA_OK;
}

int dana_batch_stats(const float* x, float* mean, float* var_biased, long rows, int channels, long ld, void* workspace,
                     size_t workspace_bytes, dana_stream_t stream) {
  DANA_CHECK_ARG(rows > 0 && channels > 0 && x && mean && var_biased, "dana_batch_stats: bad args");
  if (ld <= 0) ld = channels;
  const size_t need = dana_colsum_workspace_bytes(rows, channels);
  if (!workspace || workspace_bytes < need) {
    dana_set_error("dana_batch_stats: workspace %zu < %zu", workspace_bytes, need);
    return DANA_ERR_WORKSPACE;
  }
  const int chunks = (int)((rows + CS_ROWS - 1) / CS_ROWS);
  DANA_CHECK_ARG(chunks <= 65535, "dana_batch_stats: too many rows");
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(dana_ceil_div(channels, 64), chunks);
  const float inv = 1.f / (float)rows;
  colsum_partial_kernel<<<grid, 256, 0, s>>>(x, (float*)workspace, rows, channels, ld);
  colsum_final_kernel<<<dana_ceil_div(channels, 256), 256, 0, s>>>((const float*)workspace, mean, chunks, channels, inv, 0);
  colsqdev_partial_kernel<<<grid, 256, 0, s>>>(x, mean, (float*)workspace, rows, channels, ld);
  colsum_final_kernel<<<dana_ceil_div(channels, 256), 256, 0, s>>>((const float*)workspace, var_biased, chunks, channels,
                                                                  inv, 0);
  DANA_CHECK_LAUNCH("dana_batch_stats");
  return DANA_OK;
}

int dana_scale_shift_relu(float* x, const float* scale, const float* shift, long rows, int channels, int relu,
                          dana_stream_t stream) {
  DANA_CHECK_ARG(rows >= 0 && channels > 0 && channels % 4 == 0, "dana_scale_shift_relu: bad shape");
  if (rows == 0) return DANA_OK;
  DANA_CHECK_ARG(x && scale && shift, "dana_scale_shift_relu: null pointer");
  const long total = rows * (channels / 4);
  scale_shift_relu_kernel<<<grid_for(total, 256), 256, 0, (hipStream_t)stream>>>((float4*)x, (const float4*)scale,
                                                                                 (const float4*)shift, total,
                                                                                 channels / 4, relu);
  DANA_CHECK_LAUNCH("dana_scale_shift_relu");
  return DANA_OK;
}

int dana_avgpool_backward_nhwc(const float* grad_out, float* grad_in, int batch, int height, int width, int channels,
                               int k, int stride, dana_stream_t stream) {
  DANA_CHECK_ARG(batch >= 0 && k > 0 && stride > 0 && height >= k && width >= k && channels % 4 == 0,
                 "dana_avgpool_backward_nhwc: bad shape");
  if (batch == 0) return DANA_OK;
  DANA_CHECK_ARG(grad_out && grad_in, "dana_avgpool_backward_nhwc: null pointer");
  const int oh = (height - k) / stride + 1, ow = (width - k) / stride + 1;
  const long total = (long)batch * height * width * (channels / 4);
  avgpool_bwd_kernel<<<grid_for(total, 256), 256, 0, (hipStream_t)stream>>>((const float4*)grad_out, (float4*)grad_in,
                                                                            height, width, oh, ow, channels / 4, k, stride,
                                                                            total);
  DANA_CHECK_LAUNCH("dana_avgpool_backward_nhwc");
  return DANA_OK;
}

int dana_softmax_rows_backward(float* grad, const float* prob, long rows, int length, long ld_grad, long ld_prob,
                               dana_stream_t stream) {
  DANA_CHECK_ARG(rows >= 0 && length > 0, "dana_softmax_rows_backward: bad shape");
  if (rows == 0) return DANA_OK;
  DANA_CHECK_ARG(grad && prob, "dana_softmax_rows_backward: null pointer");
  if (ld_grad <= 0) ld_grad = length;
  if (ld_prob <= 0) ld_prob = length;
  softmax_bwd_rows_kernel<<<dana_ceil_div(rows, 4), 256, 0, (hipStream_t)stream>>>(grad, prob, rows, length, ld_grad,
                                                                                   ld_prob);
  DANA_CHECK_LAUNCH("dana_softmax_rows_backward");
  return DANA_OK;
}

int dana_gemm_small(const float* a, long a_stride_m, long a_stride_k, const float* b, long b_stride_k, long b_stride_n,
                    float* c, long c_stride_m, long c_stride_n, int m, int n, int k, float alpha, int accumulate,
                    dana_stream_t stream) {
  DANA_CHECK_ARG(m >= 0 && n >= 0 && k > 0, "dana_gemm_small: bad shape");
  if (m == 0 || n == 0) return DANA_OK;
  DANA_CHECK_ARG(a && b && c, "dana_gemm_small: null pointer");
  const long total = (long)m * n;
  if (k >= 64 && m <= 65535) {
    dim3 grid(dana_ceil_div(n, 64), m);
    gemm_small_ksplit_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(a, a_stride_m, a_stride_k, b, b_stride_k, b_stride_n, c,
                                                                   c_stride_m, c_stride_n, n, k, alpha, accumulate);
  } else {
    gemm_small_kernel<<<dana_ceil_div(total, 256), 256, 0, (hipStream_t)stream>>>(
        a, a_stride_m, a_stride_k, b, b_stride_k, b_stride_n, c, c_stride_m, c_stride_n, m, n, k, alpha, accumulate);
  }
  DANA_CHECK_LAUNCH("dana_gemm_small");
  return DANA_OK;
}

}  // extern "C"

// ---- attention adjoints ---------------------------------------------------------------------------------
namespace {

// torch.optim.Adam (train.py:84-85), one flat segment: g' = grad_scale * g + wd * p; m = b1 m + (1-b1) g';
// v = b2 v + (1-b2) g'^2; p -= lr / bc1 * m / (sqrt(v) / sqrt(bc2) + eps)
__device__ __forceinline__ void adam1(float& p, float g, float& m, float& v, float lr, float b1, float b2, float eps,
                                      float wd, float gs, float bc1, float bc2) {
  g = g * gs + wd * p;
  m = b1 * m + (1.f - b1) * g;
  v = b2 * v + (1.f - b2) * g * g;
  p -= (lr / bc1) * m / (sqrtf(v) / sqrtf(bc2) + eps);
}
__global__ void __launch_bounds__(256)
adam_kernel(float4* __restrict__ p, const float4* __restrict__ g, float4* __restrict__ m, float4* __restrict__ v, long n4,
            float lr, float b1, float b2, float eps, float wd, float gs, float bc1, float bc2) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)blockDim.x * gridDim.x) {
    float4 pv = p[i], mv = m[i], vv = v[i];
    const float4 gv = g[i];
    adam1(pv.x, gv.x, mv.x, vv.x, lr, b1, b2, eps, wd, gs, bc1, bc2);
    adam1(pv.y, gv.y, mv.y, vv.y, lr, b1, b2, eps, wd, gs, bc1, bc2);
    adam1(pv.z, gv.z, mv.z, vv.z, lr, b1, b2, eps, wd, gs, bc1, bc2);
    adam1(pv.w, gv.w, mv.w, vv.w, lr, b1, b2, eps, wd, gs, bc1, bc2);
    p[i] = pv;
    m[i] = mv;
    v[i] = vv;
  }
}

// x[i] *= s[0] with the scalar in device memory (upstream loss gradients never visit the host)
__global__ void __launch_bounds__(256) scale_by_dev_kernel(float* __restrict__ x, long n, const float* __restrict__ s) {
  const float v = s[0];
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)blockDim.x * gridDim.x) x[i] *= v;
}

// torch.optim.SGD(momentum) over a flat parameter segment (train.py:86-87): g' = g + wd * p;
// buf = first ? g' : momentum * buf + g';  p -= lr * buf.  grad_scale folds the 1/world_size of the gradient mean.
__global__ void __launch_bounds__(256)
sgd_momentum_kernel(float4* __restrict__ p, const float4* __restrict__ g, float4* __restrict__ buf, long n4, float lr,
                    float momentum, float wd, float grad_scale, int first) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)blockDim.x * gridDim.x) {
    float4 pv = p[i];
    const float4 gv = g[i];
    float4 b = first ? make_float4(0.f, 0.f, 0.f, 0.f) : buf[i];
    b.x = momentum * b.x + (gv.x * grad_scale + wd * pv.x);
    b.y = momentum * b.y + (gv.y * grad_scale + wd * pv.y);
    b.z = momentum * b.z + (gv.z * grad_scale + wd * pv.z);
    b.w = momentum * b.w + (gv.w * grad_scale + wd * pv.w);
    pv.x -= lr * b.x;
    pv.y -= lr * b.y;
    pv.z -= lr * b.z;
    pv.w -= lr * b.w;
    buf[i] = b;
    p[i] = pv;
  }
}

// adjoint of dana_rowdot (nn.Linear(dim, 1)): partial[chunk][c] = sum_r dl[r] * x[r][c]  (-> dw by colsum_final)
// and, when dx is given, dx[r][c] += dl[r] * w[c].
__global__ void __launch_bounds__(256)
rowdot_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dl, const float* __restrict__ w,
                  float* __restrict__ dx, float* __restrict__ partial, long rows, int C, long ld_x, long ld_dx) {
  __shared__ float part[4][64];
  const int col = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rl = threadIdx.x >> 6;
  const long r0 = (long)blockIdx.y * CS_ROWS, r1 = min(rows, r0 + CS_ROWS);
  float s = 0.f;
  if (col < C) {
    const float wc = w[col];
    for (long r = r0 + rl; r < r1; r += 4) {
      const float d = dl[r];
      s += d * x[r * ld_x + col];
      if (dx) dx[r * ld_dx + col] += d * wc;
    }
  }
  part[rl][threadIdx.x & 63] = s;
  __syncthreads();
  if (rl == 0 && col < C)
    partial[(long)blockIdx.y * C + col] = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
}

// out[g][p][c] (+)= alpha * in[g][c]   (adjoint of a mean / sum over p)
__global__ void __launch_bounds__(256)
broadcast_rows_kernel(const float4* __restrict__ in, float4* __restrict__ out, long groups, int P, int C4, float alpha,
                      int accumulate) {
  const long total = groups * P * C4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)blockDim.x * gridDim.x) {
    const int c = (int)(i % C4);
    const long g = i / C4 / P;
    float4 v = in[g * C4 + c];
    v.x *= alpha; v.y *= alpha; v.z *= alpha; v.w *= alpha;
    if (accumulate) {
      const float4 o = out[i];
      v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
    }
    out[i] = v;
  }
}

// BA block adjoint, second half (dana.py:133-137): with g[c] = sum_l w[l] S[l][c] and S' = S + gamma * leaky(g):
//   dg[c] = gamma * leaky'(g[c]) * colsum_l(dS')[c];   dS[l][c] = dS'[l][c] + w[l] * dg[c];   dw[l] = sum_c S[l][c] dg[c]
// one wave per row (group, l); gsum = colsum_l(dS') [groups][C], gvec = g [groups][C]
__global__ void __launch_bounds__(256)
ba_bwd_kernel(float* __restrict__ dS, const float* __restrict__ S, const float* __restrict__ wgt,
              const float* __restrict__ gvec, const float* __restrict__ gsum, float* __restrict__ dw, long rows, int L, int C,
              float gamma, float slope) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const long grp = row / L;
  const float wl = wgt[row];
  float acc = 0.f;
  for (int c = lane; c < C; c += 64) {
    const float gv = gvec[grp * C + c];
    const float dg = gamma * (gv > 0.f ? 1.f : slope) * gsum[grp * C + c];
    acc += S[row * C + c] * dg;
    dS[row * C + c] += wl * dg;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if (lane == 0) dw[row] = acc;
}

// BA block adjoint, first half: gvec[g][c] = sum_l w[g][l] S[g][l][c] and gsum[g][c] = sum_l dS'[g][l][c] for every group in
// ONE launch (round 6: it was a host loop of 4 tiny launches per group, 48 at bs 4: 0.9 ms of the replayed iteration
// with nothing beside it). Block = (group, 64 channels): 4 row groups x 64 channel lanes, fixed summation order.
__global__ void __launch_bounds__(256)
ba_bwd_prep_kernel(const float* __restrict__ S, const float* __restrict__ wgt, const float* __restrict__ dS,
                   float* __restrict__ gvec, float* __restrict__ gsum, int L, int C) {
  __shared__ float sv[4][64], ss[4][64];
  const int g = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
  float av = 0.f, as = 0.f;
  if (c < C) {
    const float* s = S + ((long)g * L) * C + c;
    const float* d = dS + ((long)g * L) * C + c;
    const float* w = wgt + (long)g * L;
    for (int l = rg; l < L; l += 4) {
      av += w[l] * s[(long)l * C];
      as += d[(long)l * C];
    }
  }
  sv[rg][threadIdx.x & 63] = av;
  ss[rg][threadIdx.x & 63] = as;
  __syncthreads();
  if (rg == 0 && c < C) {
    const int t = threadIdx.x;
    gvec[(long)g * C + c] = (sv[0][t] + sv[1][t]) + (sv[2][t] + sv[3][t]);
    gsum[(long)g * C + c] = (ss[0][t] + ss[1][t]) + (ss[2][t] + ss[3][t]);
  }
}

// A = (softmax_seg(S0) + ugamma * u) * out_scale was formed in place by attn_softmax_unary_kernel; here, in
// place on dA (same [rows][ld] shape): p = A / out_scale - ugamma * u, g = dA * out_scale,
// dS0 = alpha * p * (g - <p, g>) per segment; pad columns are zeroed. One wave per row.
__global__ void __launch_bounds__(256)
attn_softmax_unary_bwd_kernel(float* __restrict__ dA, const float* __restrict__ A, const float* __restrict__ unary,
                              long rows, long rows_per_batch, long unary_batch_stride, int nseg, int L, long ld,
                              int kpad, float ugamma, float out_scale, float alpha) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float* g = dA + row * ld;
  const float* a = A + row * ld;
  const float* u = unary + (row / rows_per_batch) * unary_batch_stride;
  const float inv = 1.f / out_scale;
  for (int sgm = 0; sgm < nseg; ++sgm) {
    float dot = 0.f;
    for (int l = lane; l < L; l += 64) {
      const float p = a[sgm * L + l] * inv - ugamma * u[sgm * L + l];
      dot += p * (g[sgm * L + l] * out_scale);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) dot += __shfl_xor(dot, o);
    for (int l = lane; l < L; l += 64) {
      const float p = a[sgm * L + l] * inv - ugamma * u[sgm * L + l];
      g[sgm * L + l] = alpha * p * (g[sgm * L + l] * out_scale - dot);
    }
  }
  for (int l = nseg * L + lane; l < kpad; l += 64) g[l] = 0.f;
}

struct AnchorGeomB {
  int A, H, W, stride, n_gt;
};
__device__ __forceinline__ float4 anchor_box_b(const float* __restrict__ base, const AnchorGeomB& g, int i) {
  const int a = i % g.A, k = i / g.A;
  const float sx = (float)((k % g.W) * g.stride), sy = (float)((k / g.W) * g.stride);
  return make_float4(base[a * 4 + 0] + sx, base[a * 4 + 1] + sy, base[a * 4 + 2] + sx, base[a * 4 + 3] + sy);
}

// d heads of the fused RPN losses (rpn_loss_kernel): d_heads[B*H*W][row stride] zero-initialised by the caller
//   cls : (softmax(s) - onehot(label)) * g_cls / count      for labels >= 0   (count read from the loss pass)
//   bbox: g_box / B * outside_w * inside_w * dSmoothL1      for labels == 1
__global__ void __launch_bounds__(256)
rpn_loss_bwd_kernel(const float* __restrict__ heads, long hs, const float* __restrict__ labels,
                    const int* __restrict__ assign, const float* __restrict__ gt, const float* __restrict__ base,
                    AnchorGeomB g, int B, float sigma, float inside_w, float outside_w,
                    const float* __restrict__ outside_w_dev, const float* __restrict__ losses3,
                    float g_cls, float g_box, const float* __restrict__ g_dev, float* __restrict__ dheads) {
  if (outside_w_dev) outside_w = outside_w_dev[0];
  if (g_dev) {  // upstream gradients of (rpn_loss_cls, rpn_loss_bbox) read on the device: no host round trip
    g_cls = g_dev[0];
    g_box = g_dev[1];
  }
  const int total = g.H * g.W * g.A;
  const long n = (long)B * total;
  const float s2 = sigma * sigma;
  const float count = losses3[2];
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)blockDim.x * gridDim.x) {
    const int b = (int)(e / total), i = (int)(e % total);
    const float label = labels[e];
    if (label < 0.f) continue;
    const int a = i % g.A, k = i / g.A;
    const long ro = ((long)b * g.H * g.W + k) * hs;
    const float s0 = heads[ro + a], s1 = heads[ro + g.A + a];
    const float m = fmaxf(s0, s1);
    const float e0 = expf(s0 - m), e1 = expf(s1 - m);
    const float p0 = e0 / (e0 + e1), p1 = e1 / (e0 + e1);
    const float sc = g_cls / count;
    dheads[ro + a] = (p0 - (label == 0.f ? 1.f : 0.f)) * sc;
    dheads[ro + g.A + a] = (p1 - (label == 1.f ? 1.f : 0.f)) * sc;
    if (label == 1.f) {
      const float4 an = anchor_box_b(base, g, i);
      const float* q = gt + ((long)b * g.n_gt + assign[e]) * 5;
      const float ew = an.z - an.x + 1.0f, eh = an.w - an.y + 1.0f;
      const float ecx = an.x + 0.5f * ew, ecy = an.y + 0.5f * eh;
      const float gw = q[2] - q[0] + 1.0f, gh = q[3] - q[1] + 1.0f;
      const float t[4] = {(q[0] + 0.5f * gw - ecx) / ew, (q[1] + 0.5f * gh - ecy) / eh, logf(gw / ew), logf(gh / eh)};
      const long bo = ro + 2 * g.A + 4 * a;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float d = inside_w * (heads[bo + j] - t[j]);
        const float ad = fabsf(d);
        const float dl = ad < 1.f / s2 ? d * s2 : (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
        dheads[bo + j] = g_box / (float)B * outside_w * inside_w * dl;
      }
    }
  }
}

}  // namespace

extern "C" {

int dana_scale_by_device_scalar(float* x, long n, const float* scalar_dev, dana_stream_t stream) {
  DANA_CHECK_ARG(n >= 0, "dana_scale_by_device_scalar: bad size");
  if (n == 0) return DANA_OK;
  DANA_CHECK_ARG(x && scalar_dev, "dana_scale_by_device_scalar: null pointer");
  scale_by_dev_kernel<<<grid_for(n, 256), 256, 0, (hipStream_t)stream>>>(x, n, scalar_dev);
  DANA_CHECK_LAUNCH("dana_scale_by_device_scalar");
  return DANA_OK;
}

int dana_adam(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long n, float lr, float beta1,
              float beta2, float eps, float weight_decay, float grad_scale, int step, dana_stream_t stream) {
  DANA_CHECK_ARG(n >= 0 && n % 4 == 0 && step >= 1, "dana_adam: n must be a multiple of 4 and step >= 1");
  if (n == 0) return DANA_OK;
  DANA_CHECK_ARG(params && grads && exp_avg && exp_avg_sq, "dana_adam: null pointer");
  const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
  adam_kernel<<<grid_for(n / 4, 256), 256, 0, (hipStream_t)stream>>>((float4*)params, (const float4*)grads,
                                                                    (float4*)exp_avg, (float4*)exp_avg_sq, n / 4, lr, beta1,
                                                                    beta2, eps, weight_decay, grad_scale, bc1, bc2);
  DANA_CHECK_LAUNCH("dana_adam");
  return DANA_OK;
}

int dana_sgd_momentum(float* params, const float* grads, float* momentum_buf, long n, float lr, float momentum,
                      float weight_decay, float grad_scale, int first_step, dana_stream_t stream) {
  DANA_CHECK_ARG(n >= 0 && n % 4 == 0, "dana_sgd_momentum: n must be a multiple of 4 (pad the flat segment)");
  if (n == 0) return DANA_OK;
  DANA_CHECK_ARG(params && grads && momentum_buf, "dana_sgd_momentum: null pointer");
  DANA_CHECK_ARG((((uintptr_t)params | (uintptr_t)grads | (uintptr_t)momentum_buf) & 15) == 0,
                 "dana_sgd_momentum: buffers must be 16-byte aligned");
  sgd_momentum_kernel<<<grid_for(n / 4, 256), 256, 0, (hipStream_t)stream>>>((float4*)params, (const float4*)grads,
                                                                            (float4*)momentum_buf, n / 4, lr, momentum,
                                                                            weight_decay, grad_scale, first_step);
  DANA_CHECK_LAUNCH("dana_sgd_momentum");
  return DANA_OK;
}

int dana_rowdot_backward(const float* x, const float* grad_out, const float* w, float* grad_x, float* grad_w, long rows,
                         int dim, long ld_x, long ld_grad_x, int accumulate_w, void* workspace, size_t workspace_bytes,
                         dana_stream_t stream) {
  DANA_CHECK_ARG(rows > 0 && dim > 0 && x && grad_out && w && grad_w, "dana_rowdot_backward: bad args");
  if (ld_x <= 0) ld_x = dim;
  if (ld_grad_x <= 0) ld_grad_x = dim;
  const size_t need = dana_colsum_workspace_bytes(rows, dim);
  if (!workspace || workspace_bytes < need) {
    dana_set_error("dana_rowdot_backward: workspace %zu < %zu", workspace_bytes, need);
    return DANA_ERR_WORKSPACE;
  }
  const int chunks = (int)((rows + CS_ROWS - 1) / CS_ROWS);
  DANA_CHECK_ARG(chunks <= 65535, "dana_rowdot_backward: too many rows");
  dim3 grid(dana_ceil_div(dim, 64), chunks);
  rowdot_bwd_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(x, grad_out, w, grad_x, (float*)workspace, rows, dim, ld_x,
                                                          ld_grad_x);
  DANA_CHECK_LAUNCH("dana_rowdot_backward(partial)");
  colsum_final_kernel<<<dana_ceil_div(dim, 256), 256, 0, (hipStream_t)stream>>>((const float*)workspace, grad_w, chunks, dim,
                                                                                1.f, accumulate_w);
  DANA_CHECK_LAUNCH("dana_rowdot_backward(final)");
  return DANA_OK;
}

int dana_broadcast_rows(const float* in, float* out, long groups, int positions, int channels, float alpha,
                        int accumulate, dana_stream_t stream) {
  DANA_CHECK_ARG(groups >= 0 && positions > 0 && channels > 0 && channels % 4 == 0, "dana_broadcast_rows: bad shape");
  if (groups == 0) return DANA_OK;
  DANA_CHECK_ARG(in && out, "dana_broadcast_rows: null pointer");
  const long total = groups * positions * (channels / 4);
  broadcast_rows_kernel<<<grid_for(total, 256), 256, 0, (hipStream_t)stream>>>((const float4*)in, (float4*)out, groups,
                                                                               positions, channels / 4, alpha, accumulate);
  DANA_CHECK_LAUNCH("dana_broadcast_rows");
  return DANA_OK;
}

int dana_ba_backward(float* grad_s, const float* s, const float* weights, const float* gvec, const float* gsum,
                     float* grad_weights, long groups, int length, int dim, float gamma, float slope,
                     dana_stream_t stream) {
  DANA_CHECK_ARG(groups >= 0 && length > 0 && dim > 0, "dana_ba_backward: bad shape");
  if (groups == 0) return DANA_OK;
  DANA_CHECK_ARG(grad_s && s && weights && gvec && gsum && grad_weights, "dana_ba_backward: null pointer");
  const long rows = groups * length;
  ba_bwd_kernel<<<dana_ceil_div(rows, 4), 256, 0, (hipStream_t)stream>>>(grad_s, s, weights, gvec, gsum, grad_weights, rows,
                                                                        length, dim, gamma, slope);
  DANA_CHECK_LAUNCH("dana_ba_backward");
  return DANA_OK;
}

int dana_ba_backward_prep(const float* s, const float* weights, const float* grad_s, float* gvec, float* gsum, long groups,
                          int length, int dim, dana_stream_t stream) {
  DANA_CHECK_ARG(groups >= 0 && groups < 65536 && length > 0 && dim > 0, "dana_ba_backward_prep: bad shape");
  if (groups == 0) return DANA_OK;
  DANA_CHECK_ARG(s && weights && grad_s && gvec && gsum, "dana_ba_backward_prep: null pointer");
  dim3 grid(dana_ceil_div(dim, 64), (unsigned)groups);
  ba_bwd_prep_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(s, weights, grad_s, gvec, gsum, length, dim);
  DANA_CHECK_LAUNCH("dana_ba_backward_prep");
  return DANA_OK;
}

int dana_attn_softmax_unary_backward(float* grad_a, const float* a, const float* unary, long rows, long rows_per_batch,
                                     long unary_batch_stride, int nseg, int length, long ld, int kpad,
                                     float unary_gamma, float out_scale, float alpha, dana_stream_t stream) {
  DANA_CHECK_ARG(rows >= 0 && rows_per_batch > 0 && nseg > 0 && length > 0 && ld >= (long)nseg * length && kpad <= ld,
                 "dana_attn_softmax_unary_backward: bad shape");
  if (rows == 0) return DANA_OK;
  DANA_CHECK_ARG(grad_a && a && unary, "dana_attn_softmax_unary_backward: null pointer");
  attn_softmax_unary_bwd_kernel<<<dana_ceil_div(rows, 4), 256, 0, (hipStream_t)stream>>>(
      grad_a, a, unary, rows, rows_per_batch, unary_batch_stride > 0 ? unary_batch_stride : (long)nseg * length, nseg,
      length, ld, kpad, unary_gamma, out_scale, alpha);
  DANA_CHECK_LAUNCH("dana_attn_softmax_unary_backward");
  return DANA_OK;
}

int dana_rpn_loss_backward(const float* heads, long head_row_stride, const float* labels, const int* argmax,
                           const float* gt_boxes, const float* base_anchors, int B, int A, int H, int W,
                           int feat_stride, int n_gt, float sigma, float inside_weight, float outside_weight,
                           const float* outside_weight_dev,
                           const float* losses3, float grad_cls, float grad_box, const float* grad_scales_dev,
                           float* grad_heads, dana_stream_t stream) {
  DANA_CHECK_ARG(B > 0 && A > 0 && H > 0 && W > 0 && n_gt > 0 && losses3, "dana_rpn_loss_backward: bad shape");
  DANA_CHECK_ARG(heads && labels && argmax && gt_boxes && base_anchors && grad_heads,
                 "dana_rpn_loss_backward: null pointer");
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(grad_heads, 0, (size_t)B * H * W * head_row_stride * sizeof(float), s) != hipSuccess) {
    dana_set_error("dana_rpn_loss_backward: memset failed");
    return DANA_ERR_HIP;
  }
  AnchorGeomB g = {A, H, W, feat_stride, n_gt};
  const long n = (long)B * H * W * A;
  int blocks = dana_ceil_div(n, 256);
  if (blocks > 2048) blocks = 2048;
  rpn_loss_bwd_kernel<<<blocks, 256, 0, s>>>(heads, head_row_stride, labels, argmax, gt_boxes, base_anchors, g, B, sigma,
                                             inside_weight, outside_weight, outside_weight_dev, losses3, grad_cls, grad_box,
                                             grad_scales_dev,
                                             grad_heads);
  DANA_CHECK_LAUNCH("dana_rpn_loss_backward");
  return DANA_OK;
}

}  // extern "C"
