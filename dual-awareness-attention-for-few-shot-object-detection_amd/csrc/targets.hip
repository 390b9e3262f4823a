// Training-target assignment for sampled RoIs on device (pulled forward because
// it sits on the critical path of the train-mode forward): the reference's _ProposalTargetLayer
// (lib/model/rpn/proposal_target_layer_cascade.py:33-213) runs ~100 tiny torch kernels, a python
// double loop with per-element indexing (:83-91) and several host syncs between the proposals and
// RoIAlign. Here it is two launches around ONE small D2H read:
//   proposal_target_prepare : IoU of every candidate (proposals + gt boxes, :43) against the image's
//       gt boxes (bbox_transform.py:212-254 incl. the zero-area masks), max / first-argmax over gt,
//       fg / bg classification (:128-133) and ascending index lists + counts via a workgroup scan.
//   [host: reads the 2*B counts and draws the SAME np.random stream as the reference (:143-175)]
//   proposal_target_gather  : picks -> rois, labels, normalised regression targets, weights (:183-204, :83-91).
// Compiled with -ffp-contract=off (the IoU rounds like the reference's unfused torch ops).
#include "common.h"
#include "../../include/dana_hip.h"

namespace {

__device__ __forceinline__ float4 cand_box(const float* __restrict__ rois, const float* __restrict__ gt, int b, int i,
                                           int n_rois, int n_gt) {
  if (i < n_rois) {
    const float* r = rois + ((long)b * n_rois + i) * 5;
    return make_float4(r[1], r[2], r[3], r[4]);
  }
  const float* g = gt + ((long)b * n_gt + (i - n_rois)) * 5;
  return make_float4(g[0], g[1], g[2], g[3]);
}

// grid = B, block = 1024. Outputs per image: max_ov[n_all], assign[n_all], fg_list/bg_list[n_all], counts[2].
__global__ void __launch_bounds__(1024)
proposal_target_prepare_kernel(const float* __restrict__ rois, const float* __restrict__ gt, int n_rois, int n_gt,
                               float fg_thresh, float bg_hi, float bg_lo, float* __restrict__ max_ov,
                               int* __restrict__ assign, int* __restrict__ fg_list, int* __restrict__ bg_list,
                               int* __restrict__ counts) {
  extern __shared__ float sgt[];  // [n_gt][6]: x1,y1,x2,y2,area,is_zero
  __shared__ int wsum_fg[16], wsum_bg[16];
  __shared__ int run_fg, run_bg;
  const int b = blockIdx.x;
  const int n_all = n_rois + n_gt;
  for (int k = threadIdx.x; k < n_gt; k += blockDim.x) {
    const float* g = gt + ((long)b * n_gt + k) * 5;
    const float gw = g[2] - g[0] + 1.f, gh = g[3] - g[1] + 1.f;
    sgt[k * 6 + 0] = g[0];
    sgt[k * 6 + 1] = g[1];
    sgt[k * 6 + 2] = g[2];
    sgt[k * 6 + 3] = g[3];
    sgt[k * 6 + 4] = gw * gh;
    sgt[k * 6 + 5] = (gw == 1.f && gh == 1.f) ? 1.f : 0.f;
  }
  if (threadIdx.x == 0) run_fg = run_bg = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i0 = 0; i0 < n_all; i0 += blockDim.x) {
    const int i = i0 + threadIdx.x;
    bool fg = false, bg = false;
    if (i < n_all) {
      const float4 a = cand_box(rois, gt, b, i, n_rois, n_gt);
      const float aw = a.z - a.x + 1.f, ah = a.w - a.y + 1.f;
      const float a_area = aw * ah;
      const bool a_zero = (aw == 1.f && ah == 1.f);
      float best = -3.4e38f;
      int arg = 0;
      for (int k = 0; k < n_gt; ++k) {
        float ov;
        if (a_zero) {
          ov = -1.f;  // masked_fill(anchors_area_zero, -1) wins over the gt mask (applied last)
        } else if (sgt[k * 6 + 5] != 0.f) {
          ov = 0.f;
        } else {
          float iw = fminf(a.z, sgt[k * 6 + 2]) - fmaxf(a.x, sgt[k * 6 + 0]) + 1.f;
          float ih = fminf(a.w, sgt[k * 6 + 3]) - fmaxf(a.y, sgt[k * 6 + 1]) + 1.f;
          iw = iw < 0.f ? 0.f : iw;
          ih = ih < 0.f ? 0.f : ih;
          const float inter = iw * ih;
          const float ua = a_area + sgt[k * 6 + 4] - inter;
          ov = inter / ua;
        }
        if (ov > best) {  // first maximum, like torch.max
          best = ov;
          arg = k;
        }
      }
      max_ov[(long)b * n_all + i] = best;
      assign[(long)b * n_all + i] = arg;
      fg = best >= fg_thresh;
      bg = (best < bg_hi) && (best >= bg_lo);
    }
    // ordered compaction: wave ballots + cross-wave prefix
    const unsigned long long mf = __ballot(fg), mb = __ballot(bg);
    if (lane == 0) {
      wsum_fg[wave] = __builtin_popcountll(mf);
      wsum_bg[wave] = __builtin_popcountll(mb);
    }
    __syncthreads();
    int off_fg = run_fg, off_bg = run_bg;
    for (int w = 0; w < wave; ++w) {
      off_fg += wsum_fg[w];
      off_bg += wsum_bg[w];
    }
    const unsigned long long below = (1ULL << lane) - 1;
    if (fg) fg_list[(long)b * n_all + off_fg + __builtin_popcountll(mf & below)] = i;
    if (bg) bg_list[(long)b * n_all + off_bg + __builtin_popcountll(mb & below)] = i;
    __syncthreads();
    if (threadIdx.x == 0) {
      int tf = 0, tb = 0;
      for (int w = 0; w < 16; ++w) {
        tf += wsum_fg[w];
        tb += wsum_bg[w];
      }
      run_fg += tf;
      run_bg += tb;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    counts[b * 2 + 0] = run_fg;
    counts[b * 2 + 1] = run_bg;
  }
}

// grid = (ceil(R/256), B). picks[b][r]: position in the fg list (r < fg_n[b]) or in the bg list.
__global__ void __launch_bounds__(256)
proposal_target_gather_kernel(const float* __restrict__ rois, const float* __restrict__ gt, int n_rois, int n_gt,
                              const int* __restrict__ assign, const int* __restrict__ fg_list,
                              const int* __restrict__ bg_list, const int* __restrict__ picks,
                              const int* __restrict__ fg_n, int R, float4 means, float4 stds, float4 inside_w,
                              int normalize, float* __restrict__ rois_out, float* __restrict__ labels_out,
                              float* __restrict__ targets, float* __restrict__ w_in, float* __restrict__ w_out) {
  const int b = blockIdx.y;
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  const int n_all = n_rois + n_gt;
  const bool is_fg = r < fg_n[b];
  const int pick = picks[b * R + r];
  // (clamped: a pick into an EMPTY list -- fg and bg both empty, which the reference refuses with a ValueError,
  // proposal_target_layer_cascade.py:186-187 -- must not turn an uninitialised list entry into a wild read)
  const int i = min(max((is_fg ? fg_list : bg_list)[(long)b * n_all + min(max(pick, 0), n_all - 1)], 0), n_all - 1);
  const float4 ex = cand_box(rois, gt, b, i, n_rois, n_gt);
  const float* g = gt + ((long)b * n_gt + assign[(long)b * n_all + i]) * 5;
  const float label = is_fg ? g[4] : 0.f;  // labels_batch[i][fg_rois_per_this_image:] = 0 (:180-181)
  float* ro = rois_out + ((long)b * R + r) * 5;
  ro[0] = (float)b;
  ro[1] = ex.x;
  ro[2] = ex.y;
  ro[3] = ex.z;
  ro[4] = ex.w;
  labels_out[(long)b * R + r] = label;
  // bbox_transform_batch (bbox_transform.py:52-69) + normalisation (:108-111)
  const float ew = ex.z - ex.x + 1.0f, eh = ex.w - ex.y + 1.0f;
  const float ecx = ex.x + 0.5f * ew, ecy = ex.y + 0.5f * eh;
  const float gw = g[2] - g[0] + 1.0f, gh = g[3] - g[1] + 1.0f;
  const float gcx = g[0] + 0.5f * gw, gcy = g[1] + 0.5f * gh;
  float t0 = (gcx - ecx) / ew, t1 = (gcy - ecy) / eh, t2 = logf(gw / ew), t3 = logf(gh / eh);
  if (normalize) {
    t0 = (t0 - means.x) / stds.x;
    t1 = (t1 - means.y) / stds.y;
    t2 = (t2 - means.z) / stds.z;
    t3 = (t3 - means.w) / stds.w;
  }
  const bool pos = label > 0.f;
  float* to = targets + ((long)b * R + r) * 4;
  float* wi = w_in + ((long)b * R + r) * 4;
  float* wo = w_out + ((long)b * R + r) * 4;
  to[0] = pos ? t0 : 0.f;
  to[1] = pos ? t1 : 0.f;
  to[2] = pos ? t2 : 0.f;
  to[3] = pos ? t3 : 0.f;
  wi[0] = pos ? inside_w.x : 0.f;
  wi[1] = pos ? inside_w.y : 0.f;
  wi[2] = pos ? inside_w.z : 0.f;
  wi[3] = pos ? inside_w.w : 0.f;
  wo[0] = (pos && inside_w.x > 0.f) ? 1.f : 0.f;
  wo[1] = (pos && inside_w.y > 0.f) ? 1.f : 0.f;
  wo[2] = (pos && inside_w.z > 0.f) ? 1.f : 0.f;
  wo[3] = (pos && inside_w.w > 0.f) ? 1.f : 0.f;
}

// ------------------------------------------------------------------------------------------------
// Anchor targets (lib/model/rpn/anchor_target_layer.py:48-193) and the RPN losses (rpn.py:97-115).
// anchor_target_prepare : per image, all H*W*A anchors in (h, w, a) order: inside-image test against
//     image 0's size (:85-86), IoU with the gt boxes, max / first-argmax over gt, per-gt max over the
//     inside anchors (LDS atomicMax on the float bits, IoU >= 0), labels before subsampling
//     (:109-124), ascending fg / bg index lists + counts.
// [host: np.random.permutation draws exactly as :137-156, needs only the counts]
// anchor_target_disable : labels[list[pos]] = -1 for the subsampled-away anchors.
// anchor_target_outputs : materialise labels / targets / weights in the reference's layouts
//     (only for callers that want `_AnchorTargetLayer`'s outputs; the model uses rpn_loss below).
// rpn_loss_kernel       : fused cross-entropy over labels >= 0 and smooth-L1 (sigma 3) over fg anchors,
//     straight from (label, argmax, anchor, gt): the four [B,4A,H,W] target tensors never exist.
struct AnchorGeom {
  int A, H, W, stride, n_gt;
};

__device__ __forceinline__ float4 anchor_box(const float* __restrict__ base, const AnchorGeom& g, int i) {
  const int a = i % g.A, k = i / g.A;
  const float sx = (float)((k % g.W) * g.stride), sy = (float)((k / g.W) * g.stride);
  return make_float4(base[a * 4 + 0] + sx, base[a * 4 + 1] + sy, base[a * 4 + 2] + sx, base[a * 4 + 3] + sy);
}

__device__ __forceinline__ float gt_iou(float4 a, float a_area, const float* __restrict__ sg) {
  if (sg[5] != 0.f) return 0.f;  // zero-area gt row (padding): masked to 0 (bbox_transform.py:212,216)
  float iw = fminf(a.z, sg[2]) - fmaxf(a.x, sg[0]) + 1.f;
  float ih = fminf(a.w, sg[3]) - fmaxf(a.y, sg[1]) + 1.f;
  iw = iw < 0.f ? 0.f : iw;
  ih = ih < 0.f ? 0.f : ih;
  const float inter = iw * ih;
  return inter / (a_area + sg[4] - inter);
}

// The assignment runs as three launches over (anchor chunk, image) so that it is not one workgroup per image (round 1:
// 290-350 us on 4 CUs, the slowest launch of the step):
//   anchor_target_iou_kernel    per-anchor max / first argmax over gt, per-gt max over the inside anchors (LDS atomicMax
//                               on the float bits, IoU >= 0, then one global atomicMax per gt and block)
//   anchor_target_label_kernel  labels before subsampling (:109-124)
//   anchor_target_lists_kernel  ascending fg / bg index lists + counts (an ordered compaction, one workgroup per 1024 anchors)
// The per-gt maxima live in the first n_gt ints of the image's fg_list row until the third launch overwrites them.
__device__ __forceinline__ void load_gt(const float* __restrict__ gt, int b, int n_gt, float* sgt) {
  for (int k = threadIdx.x; k < n_gt; k += blockDim.x) {
    const float* q = gt + ((long)b * n_gt + k) * 5;
    const float gw = q[2] - q[0] + 1.f, gh = q[3] - q[1] + 1.f;
    sgt[k * 6 + 0] = q[0];
    sgt[k * 6 + 1] = q[1];
    sgt[k * 6 + 2] = q[2];
    sgt[k * 6 + 3] = q[3];
    sgt[k * 6 + 4] = gw * gh;
    sgt[k * 6 + 5] = (gw == 1.f && gh == 1.f) ? 1.f : 0.f;
  }
}

__global__ void __launch_bounds__(256) anchor_target_zero_kernel(int* __restrict__ gmax_g, int n_gt, int total) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n_gt) gmax_g[(long)blockIdx.y * total + k] = 0;
}

// grid = (chunks, B), block = 256; dynamic LDS: sgt[n_gt][6] | gt_max bits[n_gt]
__global__ void __launch_bounds__(256)
anchor_target_iou_kernel(const float* __restrict__ gt, const float* __restrict__ im_info,
                         const float* __restrict__ base, AnchorGeom g, float* __restrict__ max_ov_out,
                         int* __restrict__ assign, int* __restrict__ gmax_g) {
  extern __shared__ float sm[];
  float* sgt = sm;
  int* gmax = (int*)(sm + g.n_gt * 6);
  const int b = blockIdx.y;
  const int total = g.H * g.W * g.A;
  load_gt(gt, b, g.n_gt, sgt);
  for (int k = threadIdx.x; k < g.n_gt; k += blockDim.x) gmax[k] = 0;  // bits of +0.0f: overlaps are >= 0
  __syncthreads();
  const float im_h = (float)(long)im_info[0], im_w = (float)(long)im_info[1];  // long(im_info[0][..]) of IMAGE 0
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < total) {
    const float4 a = anchor_box(base, g, i);
    const bool inside = a.x >= 0.f && a.y >= 0.f && a.z < im_w && a.w < im_h;
    float best = -3.4e38f;
    int arg = 0;
    if (inside) {
      const float area = (a.z - a.x + 1.f) * (a.w - a.y + 1.f);
      for (int k = 0; k < g.n_gt; ++k) {
        const float ov = gt_iou(a, area, sgt + k * 6);
        if (ov > best) {
          best = ov;
          arg = k;
        }
        // per-gt max over the inside anchors (zero-area gt rows stay at 0, like the reference's masked column)
        if (__float_as_int(ov) > gmax[k]) atomicMax(&gmax[k], __float_as_int(ov));
      }
    }
    max_ov_out[(long)b * total + i] = inside ? best : -2.f;  // -2 marks "outside the image"
    assign[(long)b * total + i] = arg;
  }
  __syncthreads();
  for (int k = threadIdx.x; k < g.n_gt; k += blockDim.x)
    if (gmax[k] > 0) atomicMax(&gmax_g[(long)b * total + k], gmax[k]);
}

// grid = (chunks, B), block = 256; dynamic LDS: sgt[n_gt][6] | gt_max[n_gt]
__global__ void __launch_bounds__(256)
anchor_target_label_kernel(const float* __restrict__ gt, const float* __restrict__ base, AnchorGeom g, float neg_ov,
                           float pos_ov, const float* __restrict__ max_ov, const int* __restrict__ gmax_g,
                           float* __restrict__ labels) {
  extern __shared__ float sm[];
  float* sgt = sm;
  float* gmax = sm + g.n_gt * 6;
  const int b = blockIdx.y;
  const int total = g.H * g.W * g.A;
  load_gt(gt, b, g.n_gt, sgt);
  for (int k = threadIdx.x; k < g.n_gt; k += blockDim.x) {
    const float m = __int_as_float(gmax_g[(long)b * total + k]);
    gmax[k] = m == 0.f ? 1e-5f : m;  // :116
  }
  __syncthreads();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const float best = max_ov[(long)b * total + i];
  float label = -1.f;
  if (best > -1.5f) {  // inside
    const float4 a = anchor_box(base, g, i);
    const float area = (a.z - a.x + 1.f) * (a.w - a.y + 1.f);
    if (best < neg_ov) label = 0.f;
    bool is_best = false;
    for (int k = 0; k < g.n_gt; ++k)
      if (sgt[k * 6 + 5] == 0.f) is_best |= (gt_iou(a, area, sgt + k * 6) == gmax[k]);
    if (is_best) label = 1.f;
    if (best >= pos_ov) label = 1.f;
  }
  labels[(long)b * total + i] = label;
}

// grid = (chunks of 1024 anchors, B), block = 256. Rounds 1-3 ran ONE 1024-thread workgroup per image: its ~5 000 wave
// instructions were issue-bound on a single CU (16 waves on 4 SIMDs: 33-35 us, however the loads were batched). Now every
// chunk is its own workgroup and there is no hand-off between them at all: a workgroup first counts the fg / bg anchors
// in FRONT of its chunk itself (a strided read of at most `total` labels, L2-resident: ~22 loads per thread for the last
// chunk of a 600 x 1000 image), then compacts its own 1024 anchors in order -- four consecutive anchors per thread, a
// wave scan of the per-thread counts, the four waves' totals through LDS. No flags, no spinning, no dependence on the
// dispatch order; the last chunk writes the image's counts.
constexpr int LISTS_CHUNK = 1024;
__global__ void __launch_bounds__(256)
anchor_target_lists_kernel(const float* __restrict__ labels, int total, int* __restrict__ fg_list,
                           int* __restrict__ bg_list, int* __restrict__ counts) {
  __shared__ int red_f[4], red_b[4], tot_f[4], tot_b[4];
  const int b = blockIdx.y, c = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* lab = labels + (long)b * total;
  // (1) anchors in front of the chunk
  int pf = 0, pb = 0;
  const int front = c * LISTS_CHUNK;
  int i = threadIdx.x;
  for (; i + 7 * 256 < front; i += 8 * 256) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = lab[i + u * 256];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      pf += v[u] == 1.f;
      pb += v[u] == 0.f;
    }
  }
  for (; i < front; i += 256) {
    const float v = lab[i];
    pf += v == 1.f;
    pb += v == 0.f;
  }
  // (2) this chunk: thread t owns anchors front + 4t .. front + 4t + 3
  const int i0 = front + 4 * threadIdx.x;
  float v[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) v[u] = (i0 + u) < total ? lab[i0 + u] : -1.f;
  int nf = 0, nb = 0;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    nf += v[u] == 1.f;
    nb += v[u] == 0.f;
  }
  // wave sums of the front counts, inclusive wave scans of the per-thread chunk counts
  int sf = nf, sb = nb;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int tf = __shfl_up(sf, o), tb = __shfl_up(sb, o);
    if (lane >= o) {
      sf += tf;
      sb += tb;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    pf += __shfl_xor(pf, o);
    pb += __shfl_xor(pb, o);
  }
  if (lane == 63) {
    tot_f[wave] = sf;
    tot_b[wave] = sb;
  }
  if (lane == 0) {
    red_f[wave] = pf;
    red_b[wave] = pb;
  }
  __syncthreads();
  int off_f = red_f[0] + red_f[1] + red_f[2] + red_f[3] + (sf - nf);
  int off_b = red_b[0] + red_b[1] + red_b[2] + red_b[3] + (sb - nb);
  for (int w = 0; w < wave; ++w) {
    off_f += tot_f[w];
    off_b += tot_b[w];
  }
  int* fl = fg_list + (long)b * total;
  int* bl = bg_list + (long)b * total;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    if (v[u] == 1.f) fl[off_f++] = i0 + u;
    if (v[u] == 0.f) bl[off_b++] = i0 + u;
  }
  if (c == (int)gridDim.x - 1 && threadIdx.x == 255) {  // (the last thread of the last chunk has seen everything)
    counts[b * 2 + 0] = off_f;
    counts[b * 2 + 1] = off_b;
  }
}

// disable[e] = (image << 1 | is_bg) packed in `which`, position in that image's list in `pos`
__global__ void __launch_bounds__(256)
anchor_target_disable_kernel(float* __restrict__ labels, const int* __restrict__ fg_list,
                             const int* __restrict__ bg_list, const int* __restrict__ which,
                             const int* __restrict__ pos, int n, int total) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  const int b = which[e] >> 1;
  const int* list = (which[e] & 1) ? bg_list : fg_list;
  labels[(long)b * total + list[(long)b * total + pos[e]]] = -1.f;
}

// the same with the entry count in device memory and (which, pos) interleaved: a captured hipGraph replays this launch
// with whatever the host drew for THIS step (the count is data, not a launch parameter)
__global__ void __launch_bounds__(256)
anchor_target_disable_dev_kernel(float* __restrict__ labels, const int* __restrict__ fg_list,
                                 const int* __restrict__ bg_list, const int* __restrict__ n_dev,
                                 const int2* __restrict__ pairs, int cap, int total) {
  const int n = min(n_dev[0], cap);
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
    const int2 wp = pairs[e];
    const int b = wp.x >> 1;
    const int* list = (wp.x & 1) ? bg_list : fg_list;
    labels[(long)b * total + list[(long)b * total + wp.y]] = -1.f;
  }
}

// bbox_transform_batch of (anchor, matched gt) (bbox_transform.py:36-75)
__device__ __forceinline__ void anchor_targets4(float4 a, const float* __restrict__ q, float t[4]) {
  const float ew = a.z - a.x + 1.0f, eh = a.w - a.y + 1.0f;
  const float ecx = a.x + 0.5f * ew, ecy = a.y + 0.5f * eh;
  const float gw = q[2] - q[0] + 1.0f, gh = q[3] - q[1] + 1.0f;
  const float gcx = q[0] + 0.5f * gw, gcy = q[1] + 0.5f * gh;
  t[0] = (gcx - ecx) / ew;
  t[1] = (gcy - ecy) / eh;
  t[2] = logf(gw / ew);
  t[3] = logf(gh / eh);
}

// outputs in the reference's layouts: labels [B][A*H*W] (a-major), targets / weights [B][4A][H*W]
__global__ void __launch_bounds__(256)
anchor_target_outputs_kernel(const float* __restrict__ labels, const float* __restrict__ max_ov,
                             const int* __restrict__ assign, const float* __restrict__ gt,
                             const float* __restrict__ base, AnchorGeom g, float inside_w, float outside_w,
                             const float* __restrict__ outside_w_dev,
                             float* __restrict__ labels_out, float* __restrict__ targets, float* __restrict__ w_in,
                             float* __restrict__ w_out) {
  if (outside_w_dev) outside_w = outside_w_dev[0];  // 1 / num_examples computed on the device (device-RNG mode)
  const int b = blockIdx.y;
  const int total = g.H * g.W * g.A, hw = g.H * g.W;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int a = i % g.A, k = i / g.A;
  const float label = labels[(long)b * total + i];
  labels_out[(long)b * total + a * hw + k] = label;
  float t[4] = {0.f, 0.f, 0.f, 0.f};
  if (max_ov[(long)b * total + i] > -1.5f)
    anchor_targets4(anchor_box(base, g, i), gt + ((long)b * g.n_gt + assign[(long)b * total + i]) * 5, t);
  const float wi = label == 1.f ? inside_w : 0.f, wo = label >= 0.f ? outside_w : 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const long o = ((long)b * 4 * g.A + a * 4 + j) * hw + k;
    targets[o] = t[j];
    w_in[o] = wi;
    w_out[o] = wo;
  }
}

// partial[block] = (sum of CE over labels >= 0, count, sum of weighted smooth-L1 over fg anchors)
__global__ void __launch_bounds__(256)
rpn_loss_kernel(const float* __restrict__ heads, long head_row_stride, const float* __restrict__ labels,
                const int* __restrict__ assign, const float* __restrict__ gt, const float* __restrict__ base,
                AnchorGeom g, int B, float sigma, float inside_w, float outside_w, const float* __restrict__ outside_w_dev,
                float* __restrict__ partial) {
  if (outside_w_dev) outside_w = outside_w_dev[0];
  __shared__ float red[3][4];
  const int total = g.H * g.W * g.A;
  const long n = (long)B * total;
  float ce = 0.f, cnt = 0.f, sl1 = 0.f;
  const float s2 = sigma * sigma;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)blockDim.x * gridDim.x) {
    const int b = (int)(e / total), i = (int)(e % total);
    const float label = labels[e];
    if (label < 0.f) continue;
    const int a = i % g.A, k = i / g.A;
    const float* row = heads + ((long)b * g.H * g.W + k) * head_row_stride;
    const float s0 = row[a], s1 = row[g.A + a];  // (bg, fg) scores of this anchor (rpn.py:47-56,98)
    const float m = fmaxf(s0, s1);
    const float lse = m + logf(expf(s0 - m) + expf(s1 - m));
    ce += lse - (label == 1.f ? s1 : s0);
    cnt += 1.f;
    if (label == 1.f) {
      float t[4];
      anchor_targets4(anchor_box(base, g, i), gt + ((long)b * g.n_gt + assign[e]) * 5, t);
      const float* bp = row + 2 * g.A + 4 * a;
#pragma unroll
      for (int j = 0; j < 4; ++j) {  // net_utils.py:71-85
        const float d = inside_w * (bp[j] - t[j]);
        const float ad = fabsf(d);
        const float l = ad < 1.f / s2 ? d * d * (s2 / 2.f) : ad - 0.5f / s2;
        sl1 += outside_w * l;
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    ce += __shfl_xor(ce, o);
    cnt += __shfl_xor(cnt, o);
    sl1 += __shfl_xor(sl1, o);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    red[0][wave] = ce;
    red[1][wave] = cnt;
    red[2][wave] = sl1;
  }
  __syncthreads();
  if (threadIdx.x < 3)
    partial[blockIdx.x * 3 + threadIdx.x] =
        (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
}

__global__ void __launch_bounds__(64)
rpn_loss_reduce_kernel(const float* __restrict__ partial, int nblocks, int B, float* __restrict__ out) {
  float ce = 0.f, cnt = 0.f, sl1 = 0.f;
  for (int i = threadIdx.x; i < nblocks; i += 64) {
    ce += partial[i * 3];
    cnt += partial[i * 3 + 1];
    sl1 += partial[i * 3 + 2];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    ce += __shfl_xor(ce, o);
    cnt += __shfl_xor(cnt, o);
    sl1 += __shfl_xor(sl1, o);
  }
  if (threadIdx.x == 0) {
    out[0] = ce / cnt;        // F.cross_entropy mean over the kept anchors (rpn.py:104)
    out[1] = sl1 / (float)B;  // sum over dims [1,2,3], mean over the batch (rpn.py:114, net_utils.py:82-84)
    out[2] = cnt;             // the cross-entropy's divisor, for dana_rpn_loss_backward
  }
}

// ---- RCNN losses (dana.py:199-217): 2-way cross-entropy with the 1:2:1 hard-negative mining + box smooth-L1 ----------
// Rows 0..n-1 are the positive-support head's scores (labels from the proposal-target layer), rows n..2n-1 the
// negative-support head's (labels 0). fg = label 1; the background rows are ranked by their foreground probability
// (descending; ties by row index) separately in each half, the first bg_num_0 of the first half and the first bg_num_1
// of the second are kept (dana.py:204-213), and the cross-entropy is the mean over fg + kept rows.
//   phase A  per row: p1 = softmax(score)[1], ce, fg flag; per block: fg count, smooth-L1 partial
//   phase B  per row: rank among the background rows of its half -> selected?; per block: CE sum / count partials
//   phase C  reduce -> losses; scale the gradient seeds by 1 / count
struct RcnnLossWs {
  float* p1;       // [2n]
  float* ce;       // [2n]
  float* partial;  // [blocks][4]: fg count, smooth-L1 sum, ce sum, selected count
};

__global__ void __launch_bounds__(256)
rcnn_loss_a_kernel(const float* __restrict__ sp, const float* __restrict__ sn, const float* __restrict__ labels,
                   const float* __restrict__ pred, const float* __restrict__ tgt, const float* __restrict__ w_in,
                   const float* __restrict__ w_out, int n, float sigma, RcnnLossWs ws, float* __restrict__ gbox) {
  __shared__ float red[2][4];
  const int r = blockIdx.x * 256 + threadIdx.x;
  float fg = 0.f, sl1 = 0.f;
  if (r < 2 * n) {
    const float* s = r < n ? sp + (long)r * 2 : sn + (long)(r - n) * 2;
    const float lab = r < n ? labels[r] : 0.f;
    const float m = fmaxf(s[0], s[1]);
    const float e0 = expf(s[0] - m), e1 = expf(s[1] - m);
    ws.p1[r] = e1 / (e0 + e1);
    ws.ce[r] = (m + logf(e0 + e1)) - (lab == 1.f ? s[1] : s[0]);
    fg = lab == 1.f ? 1.f : 0.f;
    if (r < n) {  // net_utils.py:71-85 with sigma, dim=[1]; the mean over rows is applied in phase C
      const float s2 = sigma * sigma;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float d = w_in[r * 4 + j] * (pred[r * 4 + j] - tgt[r * 4 + j]);
        const float ad = fabsf(d);
        sl1 += w_out[r * 4 + j] * (ad < 1.f / s2 ? d * d * (s2 / 2.f) : ad - 0.5f / s2);
        if (gbox) {
          const float dl = ad < 1.f / s2 ? d * s2 : (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
          gbox[r * 4 + j] = w_out[r * 4 + j] * w_in[r * 4 + j] * dl / (float)n;
        }
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    fg += __shfl_xor(fg, o);
    sl1 += __shfl_xor(sl1, o);
  }
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = fg;
    red[1][threadIdx.x >> 6] = sl1;
  }
  __syncthreads();
  if (threadIdx.x < 2)
    ws.partial[blockIdx.x * 4 + threadIdx.x] =
        (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
}

__global__ void __launch_bounds__(256)
rcnn_loss_b_kernel(const float* __restrict__ sp, const float* __restrict__ sn, const float* __restrict__ labels, int n,
                   int blocks, RcnnLossWs ws, float* __restrict__ gsp, float* __restrict__ gsn, int use_lds) {
  __shared__ float red[2][4];
  __shared__ int s_fg;
  if (threadIdx.x == 0) {
    float c = 0.f;
    for (int i = 0; i < blocks; ++i) c += ws.partial[i * 4];
    s_fg = (int)c;
  }
  // the rank loop below reads every background probability of the row's half: staged in LDS once per workgroup (a row
  // that is not background gets -inf, so the loop needs no label test); from global memory the 2n dependent-latency
  // iterations made this kernel 115 us of the step's serial tail
  extern __shared__ float s_p1[];  // [2n] when use_lds
  const int n_all = 2 * n;
  if (use_lds)
    for (int j = threadIdx.x; j < n_all; j += 256) s_p1[j] = (j < n ? labels[j] : 0.f) == 0.f ? ws.p1[j] : -INFINITY;
  __syncthreads();
  const int nfg = s_fg;
  const int bg0 = max(1, min(nfg * 2, (int)(n_all * 0.25)));  // dana.py:207-208
  const int bg1 = max(1, min(nfg, bg0));
  const int r = blockIdx.x * 256 + threadIdx.x;
  float ce = 0.f, cnt = 0.f;
  if (r < n_all) {
    const float lab = r < n ? labels[r] : 0.f;
    bool sel = lab == 1.f;
    if (lab == 0.f) {
      const int h0 = r < n ? 0 : n, h1 = r < n ? n : n_all;
      const float mine = ws.p1[r];
      int rank = 0;
      if (use_lds) {
        for (int j = h0; j < h1; ++j) {
          const float pj = s_p1[j];
          rank += (pj > mine || (pj == mine && j < r)) ? 1 : 0;
        }
      } else {
        for (int j = h0; j < h1; ++j) {
          const float lj = j < n ? labels[j] : 0.f;
          const float pj = ws.p1[j];
          rank += (lj == 0.f && (pj > mine || (pj == mine && j < r))) ? 1 : 0;
        }
      }
      sel = rank < (r < n ? bg0 : bg1);
    }
    if (sel) {
      ce = ws.ce[r];
      cnt = 1.f;
    }
    float* g = r < n ? (gsp ? gsp + (long)r * 2 : nullptr) : (gsn ? gsn + (long)(r - n) * 2 : nullptr);
    if (g) {  // unscaled seed (softmax - onehot) on kept rows; phase C divides by the count
      const float p1 = ws.p1[r];
      g[0] = sel ? (1.f - p1) - (lab == 1.f ? 0.f : 1.f) : 0.f;
      g[1] = sel ? p1 - (lab == 1.f ? 1.f : 0.f) : 0.f;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    ce += __shfl_xor(ce, o);
    cnt += __shfl_xor(cnt, o);
  }
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = ce;
    red[1][threadIdx.x >> 6] = cnt;
  }
  __syncthreads();
  if (threadIdx.x < 2)
    ws.partial[blockIdx.x * 4 + 2 + threadIdx.x] =
        (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
}

__global__ void __launch_bounds__(256)
rcnn_loss_c_kernel(int n, int blocks, RcnnLossWs ws, float* __restrict__ losses, float* __restrict__ gsp,
                   float* __restrict__ gsn) {
  float sl1 = 0.f, ce = 0.f, cnt = 0.f;
  for (int i = 0; i < blocks; ++i) {  // fixed order: deterministic
    sl1 += ws.partial[i * 4 + 1];
    ce += ws.partial[i * 4 + 2];
    cnt += ws.partial[i * 4 + 3];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    losses[0] = ce / cnt;        // F.cross_entropy mean over the kept rows (dana.py:215)
    losses[1] = sl1 / (float)n;  // _smooth_l1_loss(...).mean() over the rois (dana.py:217)
    losses[2] = cnt;
  }
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r < 2 * n) {
    float* g = r < n ? (gsp ? gsp + (long)r * 2 : nullptr) : (gsn ? gsn + (long)(r - n) * 2 : nullptr);
    if (g) {  // no kept row: the loss is 0/0 = NaN like F.cross_entropy of an empty selection, its gradient is zero
      g[0] = cnt > 0.f ? g[0] / cnt : 0.f;
      g[1] = cnt > 0.f ? g[1] / cnt : 0.f;
    }
  }
}

// Plain classification + box-regression losses of the sibling detectors (faster_rcnn.py:93-98): F.cross_entropy over all n
// rois (mean) and _smooth_l1_loss(sigma) (net_utils.py:71-85: sum over the 4 coordinates, mean over rois), with their
// gradient seeds. ONE workgroup: n is a few hundred rows; the sums go through LDS in a fixed order (deterministic).
__global__ void __launch_bounds__(256)
plain_rcnn_loss_kernel(const float* __restrict__ scores, const long long* __restrict__ labels, const float* __restrict__ bbox,
                       const float* __restrict__ tgt, const float* __restrict__ win, const float* __restrict__ wout, int n,
                       int C, float sigma, float* __restrict__ losses2, float* __restrict__ gscores, float* __restrict__ gbbox) {
  __shared__ float part[2][256];
  const float s2 = sigma * sigma, inv_n = 1.f / (float)n;
  float lc = 0.f, lb = 0.f;
  for (int r = threadIdx.x; r < n; r += 256) {
    const float* sr = scores + (long)r * C;
    float m = sr[0];
    for (int c = 1; c < C; ++c) m = fmaxf(m, sr[c]);
    float z = 0.f;
    for (int c = 0; c < C; ++c) z += expf(sr[c] - m);
    // a label outside 0..C-1 (class count and score width disagree) must not index the score row: the loss and that
    // row's gradient become NaN -- loud in the very next optimizer step -- where F.cross_entropy would raise; there is
    // no ignore_index on this path (faster_rcnn.py:94 passes every sampled roi)
    const long long yl = labels[r];
    const bool ok = yl >= 0 && yl < (long long)C;
    const int y = ok ? (int)yl : 0;
    lc += ok ? -(sr[y] - m - logf(z)) : __builtin_nanf("");
    if (gscores)
      for (int c = 0; c < C; ++c)
        gscores[(long)r * C + c] = ok ? (expf(sr[c] - m) / z - (c == y ? 1.f : 0.f)) * inv_n : __builtin_nanf("");
    float row = 0.f;
    for (int q = 0; q < 4; ++q) {
      const long i = (long)r * 4 + q;
      const float d = win[i] * (bbox[i] - tgt[i]);
      const float a = fabsf(d);
      const bool near = a < 1.f / s2;
      row += wout[i] * (near ? d * d * (s2 * 0.5f) : a - 0.5f / s2);
      if (gbbox) gbbox[i] = wout[i] * win[i] * (near ? s2 * d : (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f))) * inv_n;
    }
    lb += row;
  }
  part[0][threadIdx.x] = lc;
  part[1][threadIdx.x] = lb;
  __syncthreads();
  if (threadIdx.x < 2) {
    float t = 0.f;
    for (int i = 0; i < 256; ++i) t += part[threadIdx.x][i];
    losses2[threadIdx.x] = t * inv_n;
  }
}

}  // namespace

extern "C" {

int dana_plain_rcnn_loss(const float* scores, const long long* labels, const float* bbox_pred, const float* bbox_targets,
                         const float* inside_weights, const float* outside_weights, int n, int n_classes, float sigma,
                         float* losses2, float* grad_scores, float* grad_bbox, dana_stream_t stream) {
  DANA_CHECK_ARG(n > 0 && n_classes > 0, "dana_plain_rcnn_loss: bad shape");
  DANA_CHECK_ARG(scores && labels && bbox_pred && bbox_targets && inside_weights && outside_weights && losses2,
                 "dana_plain_rcnn_loss: null pointer");
  plain_rcnn_loss_kernel<<<1, 256, 0, (hipStream_t)stream>>>(scores, labels, bbox_pred, bbox_targets, inside_weights,
                                                             outside_weights, n, n_classes, sigma, losses2, grad_scores, grad_bbox);
  DANA_CHECK_LAUNCH("dana_plain_rcnn_loss");
  return DANA_OK;
}

size_t dana_rcnn_loss_workspace_bytes(int n) {
  if (n <= 0) return 0;
  const int blocks = (2 * n + 255) / 256;
  return ((size_t)4 * n + (size_t)blocks * 4) * sizeof(float);
}

int dana_rcnn_loss(const float* score_pos, const float* score_neg, const float* labels, const float* bbox_pred,
                   const float* bbox_targets, const float* inside_weights, const float* outside_weights, int n,
                   float sigma, float* losses3, float* grad_score_pos, float* grad_score_neg, float* grad_bbox,
                   void* workspace, size_t workspace_bytes, dana_stream_t stream) {
  DANA_CHECK_ARG(n > 0, "dana_rcnn_loss: bad shape");
  DANA_CHECK_ARG(score_pos && score_neg && labels && bbox_pred && bbox_targets && inside_weights && outside_weights &&
                     losses3,
                 "dana_rcnn_loss: null pointer");
  if (!workspace || workspace_bytes < dana_rcnn_loss_workspace_bytes(n)) {
    dana_set_error("dana_rcnn_loss: workspace too small");
    return DANA_ERR_WORKSPACE;
  }
  const int blocks = (2 * n + 255) / 256;
  RcnnLossWs ws;
  ws.p1 = (float*)workspace;
  ws.ce = ws.p1 + 2 * n;
  ws.partial = ws.ce + 2 * n;
  hipStream_t s = (hipStream_t)stream;
  rcnn_loss_a_kernel<<<blocks, 256, 0, s>>>(score_pos, score_neg, labels, bbox_pred, bbox_targets, inside_weights,
                                            outside_weights, n, sigma, ws, grad_bbox);
  DANA_CHECK_LAUNCH("dana_rcnn_loss(a)");
  const size_t b_lds = (size_t)2 * n * sizeof(float);
  const int use_lds = b_lds <= 48 * 1024;
  rcnn_loss_b_kernel<<<blocks, 256, use_lds ? b_lds : 0, s>>>(score_pos, score_neg, labels, n, blocks, ws, grad_score_pos,
                                                             grad_score_neg, use_lds);
  DANA_CHECK_LAUNCH("dana_rcnn_loss(b)");
  rcnn_loss_c_kernel<<<blocks, 256, 0, s>>>(n, blocks, ws, losses3, grad_score_pos, grad_score_neg);
  DANA_CHECK_LAUNCH("dana_rcnn_loss(c)");
  return DANA_OK;
}

int dana_proposal_target_prepare(const float* rois, const float* gt_boxes, int B, int n_rois, int n_gt,
                                 float fg_thresh, float bg_thresh_hi, float bg_thresh_lo, float* max_overlaps,
                                 int* gt_assignment, int* fg_list, int* bg_list, int* counts, dana_stream_t stream) {
  DANA_CHECK_ARG(B >= 0 && n_rois >= 0 && n_gt > 0, "dana_proposal_target_prepare: bad shape");
  if (B == 0) return DANA_OK;
  DANA_CHECK_ARG(rois && gt_boxes && max_overlaps && gt_assignment && fg_list && bg_list && counts,
                 "dana_proposal_target_prepare: null pointer");
  DANA_CHECK_ARG((size_t)n_gt * 24 <= 48 * 1024, "dana_proposal_target_prepare: too many gt boxes");
  proposal_target_prepare_kernel<<<B, 1024, (size_t)n_gt * 6 * sizeof(float), (hipStream_t)stream>>>(
      rois, gt_boxes, n_rois, n_gt, fg_thresh, bg_thresh_hi, bg_thresh_lo, max_overlaps, gt_assignment, fg_list,
      bg_list, counts);
  DANA_CHECK_LAUNCH("dana_proposal_target_prepare");
  return DANA_OK;
}

int dana_proposal_target_gather(const float* rois, const float* gt_boxes, int B, int n_rois, int n_gt,
                                const int* gt_assignment, const int* fg_list, const int* bg_list, const int* picks,
                                const int* fg_taken, int rois_per_image, const float* means4, const float* stds4,
                                const float* inside_w4, int normalize, float* rois_out, float* labels_out,
                                float* bbox_targets, float* inside_weights, float* outside_weights,
                                dana_stream_t stream) {
  DANA_CHECK_ARG(B >= 0 && n_rois >= 0 && n_gt > 0 && rois_per_image > 0, "dana_proposal_target_gather: bad shape");
  if (B == 0) return DANA_OK;
  DANA_CHECK_ARG(rois && gt_boxes && gt_assignment && fg_list && bg_list && picks && fg_taken && means4 && stds4 &&
                     inside_w4 && rois_out && labels_out && bbox_targets && inside_weights && outside_weights,
                 "dana_proposal_target_gather: null pointer");
  // means/stds/inside weights are 4 HOST floats (cfg.TRAIN.BBOX_NORMALIZE_*, BBOX_INSIDE_WEIGHTS)
  const float4 m = make_float4(means4[0], means4[1], means4[2], means4[3]);
  const float4 sd = make_float4(stds4[0], stds4[1], stds4[2], stds4[3]);
  const float4 iw = make_float4(inside_w4[0], inside_w4[1], inside_w4[2], inside_w4[3]);
  dim3 grid(dana_ceil_div(rois_per_image, 256), B);
  proposal_target_gather_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(
      rois, gt_boxes, n_rois, n_gt, gt_assignment, fg_list, bg_list, picks, fg_taken, rois_per_image, m, sd, iw,
      normalize, rois_out, labels_out, bbox_targets, inside_weights, outside_weights);
  DANA_CHECK_LAUNCH("dana_proposal_target_gather");
  return DANA_OK;
}

int dana_anchor_target_prepare(const float* gt_boxes, const float* im_info, const float* base_anchors, int B,
                               int A, int H, int W, int feat_stride, int n_gt, float negative_overlap,
                               float positive_overlap, float* labels, float* max_overlaps, int* argmax,
                               int* fg_list, int* bg_list, int* counts, dana_stream_t stream) {
  DANA_CHECK_ARG(B >= 0 && A > 0 && H > 0 && W > 0 && n_gt > 0, "dana_anchor_target_prepare: bad shape");
  if (B == 0) return DANA_OK;
  DANA_CHECK_ARG(gt_boxes && im_info && base_anchors && labels && max_overlaps && argmax && fg_list && bg_list &&
                     counts,
                 "dana_anchor_target_prepare: null pointer");
  DANA_CHECK_ARG((size_t)n_gt * 28 <= 48 * 1024, "dana_anchor_target_prepare: too many gt boxes");
  AnchorGeom g = {A, H, W, feat_stride, n_gt};
  const int total = A * H * W;
  DANA_CHECK_ARG(n_gt <= total, "dana_anchor_target_prepare: more gt boxes than anchors");
  hipStream_t s = (hipStream_t)stream;
  // per-gt maxima: the first n_gt ints of every image's fg_list row, zeroed here (bits of +0.0f) by a kernel -- a
  // strided (2-D) memset node did not replay reliably inside a captured hipGraph on ROCm 7.2
  anchor_target_zero_kernel<<<dim3(dana_ceil_div(n_gt, 256), B), 256, 0, s>>>(fg_list, n_gt, total);
  const dim3 grid(dana_ceil_div(total, 256), B);
  const size_t lds = (size_t)n_gt * 7 * sizeof(float);
  anchor_target_iou_kernel<<<grid, 256, lds, s>>>(gt_boxes, im_info, base_anchors, g, max_overlaps, argmax, fg_list);
  anchor_target_label_kernel<<<grid, 256, lds, s>>>(gt_boxes, base_anchors, g, negative_overlap, positive_overlap,
                                                    max_overlaps, fg_list, labels);
  anchor_target_lists_kernel<<<dim3(dana_ceil_div(total, LISTS_CHUNK), B), 256, 0, s>>>(labels, total, fg_list, bg_list, counts);
  DANA_CHECK_LAUNCH("dana_anchor_target_prepare");
  return DANA_OK;
}

int dana_anchor_target_disable(float* labels, const int* fg_list, const int* bg_list, const int* which,
                               const int* pos, int n, int anchors_per_image, dana_stream_t stream) {
  DANA_CHECK_ARG(n >= 0 && anchors_per_image > 0, "dana_anchor_target_disable: bad shape");
  if (n == 0) return DANA_OK;
  DANA_CHECK_ARG(labels && fg_list && bg_list && which && pos, "dana_anchor_target_disable: null pointer");
  anchor_target_disable_kernel<<<dana_ceil_div(n, 256), 256, 0, (hipStream_t)stream>>>(labels, fg_list, bg_list,
                                                                                    which, pos, n, anchors_per_image);
  DANA_CHECK_LAUNCH("dana_anchor_target_disable");
  return DANA_OK;
}

int dana_anchor_target_disable_dev(float* labels, const int* fg_list, const int* bg_list, const int* n_dev,
                                   const int* which_pos_pairs, int capacity, int anchors_per_image,
                                   dana_stream_t stream) {
  DANA_CHECK_ARG(capacity >= 0 && anchors_per_image > 0, "dana_anchor_target_disable_dev: bad shape");
  if (capacity == 0) return DANA_OK;
  DANA_CHECK_ARG(labels && fg_list && bg_list && n_dev && which_pos_pairs, "dana_anchor_target_disable_dev: null pointer");
  DANA_CHECK_ARG(((uintptr_t)which_pos_pairs & 7) == 0, "dana_anchor_target_disable_dev: pairs must be 8-byte aligned");
  const int blocks = dana_ceil_div(capacity, 256) < 1024 ? dana_ceil_div(capacity, 256) : 1024;
  anchor_target_disable_dev_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(
      labels, fg_list, bg_list, n_dev, (const int2*)which_pos_pairs, capacity, anchors_per_image);
  DANA_CHECK_LAUNCH("dana_anchor_target_disable_dev");
  return DANA_OK;
}

int dana_anchor_target_outputs(const float* labels, const float* max_overlaps, const int* argmax,
                               const float* gt_boxes, const float* base_anchors, int B, int A, int H, int W,
                               int feat_stride, int n_gt, float inside_weight, float outside_weight,
                               const float* outside_weight_dev,
                               float* labels_out, float* bbox_targets, float* inside_weights,
                               float* outside_weights, dana_stream_t stream) {
  DANA_CHECK_ARG(B >= 0 && A > 0 && H > 0 && W > 0 && n_gt > 0, "dana_anchor_target_outputs: bad shape");
  if (B == 0) return DANA_OK;
  DANA_CHECK_ARG(labels && max_overlaps && argmax && gt_boxes && base_anchors && labels_out && bbox_targets &&
                     inside_weights && outside_weights,
                 "dana_anchor_target_outputs: null pointer");
  AnchorGeom g = {A, H, W, feat_stride, n_gt};
  dim3 grid(dana_ceil_div((long)H * W * A, 256), B);
  anchor_target_outputs_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(labels, max_overlaps, argmax, gt_boxes,
                                                                      base_anchors, g, inside_weight, outside_weight, outside_weight_dev,
                                                                      labels_out, bbox_targets, inside_weights,
                                                                      outside_weights);
  DANA_CHECK_LAUNCH("dana_anchor_target_outputs");
  return DANA_OK;
}

size_t dana_rpn_loss_workspace_bytes(void) { return 512 * 3 * sizeof(float); }

int dana_rpn_loss(const float* heads, long head_row_stride, const float* labels, const int* argmax,
                  const float* gt_boxes, const float* base_anchors, int B, int A, int H, int W, int feat_stride,
                  int n_gt, float sigma, float inside_weight, float outside_weight, const float* outside_weight_dev,
                  float* losses3, void* workspace,
                  size_t workspace_bytes, dana_stream_t stream) {
  DANA_CHECK_ARG(B > 0 && A > 0 && H > 0 && W > 0 && n_gt > 0 && head_row_stride >= 6 * A, "dana_rpn_loss: bad shape");
  DANA_CHECK_ARG(heads && labels && argmax && gt_boxes && base_anchors && losses3, "dana_rpn_loss: null pointer");
  if (!workspace || workspace_bytes < dana_rpn_loss_workspace_bytes()) {
    dana_set_error("dana_rpn_loss: workspace too small");
    return DANA_ERR_WORKSPACE;
  }
  AnchorGeom g = {A, H, W, feat_stride, n_gt};
  const long n = (long)B * H * W * A;
  int blocks = dana_ceil_div(n, 256);
  if (blocks > 512) blocks = 512;
  rpn_loss_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(heads, head_row_stride, labels, argmax, gt_boxes,
                                                           base_anchors, g, B, sigma, inside_weight, outside_weight, outside_weight_dev,
                                                           (float*)workspace);
  DANA_CHECK_LAUNCH("dana_rpn_loss");
  rpn_loss_reduce_kernel<<<1, 64, 0, (hipStream_t)stream>>>((const float*)workspace, blocks, B, losses3);
  DANA_CHECK_LAUNCH("dana_rpn_loss(reduce)");
  return DANA_OK;
}

}  // extern "C"
