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
  const int i = (is_fg ? fg_list : bg_list)[(long)b * n_all + pick];
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

}  // namespace

extern "C" {

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

}  // extern "C"
