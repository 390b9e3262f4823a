// RPN proposal path on device: anchor grid + delta decode + clip + fg score, per-image
// descending sort, top-N gather, batched NMS, zero-padded roi assembly.
//
// Replaces (semantics, not code) the reference's
//   lib/model/rpn/proposal_layer.py:49-190   (_ProposalLayer.forward)
//   lib/model/rpn/bbox_transform.py:77-103   (bbox_transform_inv), :125-133 (clip_boxes)
//   lib/model/rpn/rpn.py:67-69               (2-way softmax of the bg/fg score pair)
// which run as ~15 small torch kernels, a host numpy anchor grid and a python loop over the
// batch with a blocking NMS per image. Here: 4 launches for the whole batch, no host sync.
//
// Compiled with -ffp-contract=off (decode rounds like the reference's unfused torch ops).
#include "common.h"
#include <atomic>
#include "../../include/dana_hip_debug.h"

namespace {

std::atomic<int> g_sort_mode{0};  // dana_set_sort_mode: 0 = measured dispatch, 1 = the sample sort for every row, 2 = the single-workgroup kernel wherever it can run

// one lane per (image, cell k=h*W+w, anchor a); output index i = k*A + a (proposal_layer.py:98-103)
__global__ void __launch_bounds__(256)
rpn_decode_kernel(const float* __restrict__ cls, long cls_sb, long cls_sc, long cls_sp, int cls_is_prob,
                  const float* __restrict__ bbox, long bbox_sb, long bbox_sc, long bbox_sp,
                  const float* __restrict__ im_info, const float* __restrict__ base_anchors, int A, int H, int W,
                  int feat_stride, float4* __restrict__ proposals, float* __restrict__ scores) {
  const int b = blockIdx.y;
  const int n = H * W * A;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int a = i % A, k = i / A;
  const int w = k % W, h = k / W;
  // fg probability: channel A+a (proposal_layer.py:67); softmax over the (bg=a, fg=A+a) pair (rpn.py:67-69)
  const float* cp = cls + b * cls_sb + k * cls_sp;
  float fg = cp[(long)(A + a) * cls_sc];
  if (!cls_is_prob) {
    float bg = cp[(long)a * cls_sc];
    float m = fmaxf(bg, fg);
    float e0 = expf(bg - m), e1 = expf(fg - m);
    fg = e1 / (e0 + e1);
  }
  const float* bp = bbox + b * bbox_sb + k * bbox_sp + (long)(4 * a) * bbox_sc;
  const float dx = bp[0], dy = bp[bbox_sc], dw = bp[2 * bbox_sc], dh = bp[3 * bbox_sc];
  const float sx = (float)(w * feat_stride), sy = (float)(h * feat_stride);
  const float ax1 = base_anchors[a * 4 + 0] + sx, ay1 = base_anchors[a * 4 + 1] + sy;
  const float ax2 = base_anchors[a * 4 + 2] + sx, ay2 = base_anchors[a * 4 + 3] + sy;
  const float widths = ax2 - ax1 + 1.0f, heights = ay2 - ay1 + 1.0f;
  const float ctr_x = ax1 + 0.5f * widths, ctr_y = ay1 + 0.5f * heights;
  const float pcx = dx * widths + ctr_x, pcy = dy * heights + ctr_y;
  const float pw = expf(dw) * widths, ph = expf(dh) * heights;  // unclamped exp (bbox_transform.py:90-91)
  const float im_h = im_info[b * 3 + 0], im_w = im_info[b * 3 + 1];
  float x1 = pcx - 0.5f * pw, y1 = pcy - 0.5f * ph, x2 = pcx + 0.5f * pw, y2 = pcy + 0.5f * ph;
  x1 = fminf(fmaxf(x1, 0.f), im_w - 1.f);
  y1 = fminf(fmaxf(y1, 0.f), im_h - 1.f);
  x2 = fminf(fmaxf(x2, 0.f), im_w - 1.f);
  y2 = fminf(fmaxf(y2, 0.f), im_h - 1.f);
  proposals[(long)b * n + i] = make_float4(x1, y1, x2, y2);
  scores[(long)b * n + i] = fg;
}

// Sort keys: one 64-bit key per (image, anchor) = (B-1-image) << 32 | monotone(score), so ONE
// device-wide descending radix sort orders every image's scores at once (image ascending, score
// descending, ties in index order because the sort is stable) -- a segmented sort of 4 x 28 728 keys
// runs one workgroup per segment and took 0.7 ms; this takes a few 10 us.
__device__ __forceinline__ unsigned monotone_bits(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float from_monotone_bits(unsigned m) {
  return __uint_as_float((m & 0x80000000u) ? (m & 0x7FFFFFFFu) : ~m);
}
__global__ void __launch_bounds__(256)
gather_boxes_kernel(const float4* __restrict__ src, const int* __restrict__ order, int n, int order_stride, int topn,
                    float4* __restrict__ dst) {
  const int b = blockIdx.y;
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= topn) return;
  dst[(long)b * topn + r] = src[(long)b * n + order[(long)b * order_stride + r]];
}

// rois[b][r] = (b, box[keep[r]]) for r < num_keep[b], (b,0,0,0,0) after (proposal_layer.py:186-188)
__global__ void __launch_bounds__(256)
rois_assemble_kernel(const float4* __restrict__ boxes, const int* __restrict__ keep, const int* __restrict__ num_keep,
                     int topn, int keep_stride, int post_n, float* __restrict__ rois) {
  const int b = blockIdx.y;
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= post_n) return;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (r < num_keep[b]) v = boxes[(long)b * topn + keep[(long)b * keep_stride + r]];
  float* o = rois + ((long)b * post_n + r) * 5;
  o[0] = (float)b;
  o[1] = v.x;
  o[2] = v.y;
  o[3] = v.z;
  o[4] = v.w;
}

__global__ void __launch_bounds__(256)
dets_assemble_kernel(const float4* __restrict__ boxes, const float* __restrict__ scores, int R, float* __restrict__ dets) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  const float4 b = boxes[r];
  float* o = dets + (long)r * 5;
  o[0] = b.x;
  o[1] = b.y;
  o[2] = b.z;
  o[3] = b.w;
  o[4] = scores[r];
}

// inference.py:106-125 for one image: de-normalise the deltas (x stds + means), bbox_transform_inv on the
// rois, clip to the image, divide by the image scale; score = cls_prob[:, 1], masked to -inf when it does
// not pass `thresh` so that the descending sort puts every discarded row after the kept ones.
__global__ void __launch_bounds__(256)
detect_decode_kernel(const float* __restrict__ rois, const float* __restrict__ cls_prob,
                     const float* __restrict__ bbox_pred, const float* __restrict__ im_info, int R, float4 stds,
                     float4 means, int normalize, float thresh, float4* __restrict__ boxes, float* __restrict__ scores,
                     int* __restrict__ count) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  bool valid = false;
  if (r < R) {
    const float* q = rois + (long)r * 5;
    float dx = bbox_pred[r * 4 + 0], dy = bbox_pred[r * 4 + 1], dw = bbox_pred[r * 4 + 2], dh = bbox_pred[r * 4 + 3];
    if (normalize) {
      dx = dx * stds.x + means.x;
      dy = dy * stds.y + means.y;
      dw = dw * stds.z + means.z;
      dh = dh * stds.w + means.w;
    }
    const float widths = q[3] - q[1] + 1.0f, heights = q[4] - q[2] + 1.0f;
    const float ctr_x = q[1] + 0.5f * widths, ctr_y = q[2] + 0.5f * heights;
    const float pcx = dx * widths + ctr_x, pcy = dy * heights + ctr_y;
    const float pw = expf(dw) * widths, ph = expf(dh) * heights;
    const float im_h = im_info[0], im_w = im_info[1], scale = im_info[2];
    float x1 = fminf(fmaxf(pcx - 0.5f * pw, 0.f), im_w - 1.f), y1 = fminf(fmaxf(pcy - 0.5f * ph, 0.f), im_h - 1.f);
    float x2 = fminf(fmaxf(pcx + 0.5f * pw, 0.f), im_w - 1.f), y2 = fminf(fmaxf(pcy + 0.5f * ph, 0.f), im_h - 1.f);
    boxes[r] = make_float4(x1 / scale, y1 / scale, x2 / scale, y2 / scale);
    const float sc = cls_prob[r * 2 + 1];
    valid = sc > thresh;
    scores[r] = valid ? sc : -__builtin_huge_valf();
  }
  const unsigned long long m = __ballot(valid);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(count, __builtin_popcountll(m));
}

// ---- hand-written top-k + sort (round 4): proposal_layer.py:135-150 sorts every anchor's score and keeps the first
// pre_nms_topN. One 1024-lane workgroup per image does exactly that and nothing more:
//   1. keys ~monotone(score) (ascending = descending score) live in REGISTERS, up to 40 per lane;
//   2. radix select, 4 passes of 8 bits from the top: the key T of rank topn and how many of its ties belong to the top
//      (histograms in LDS, one atomic per distinct digit and wave: lanes with equal digits are matched with 8 ballots);
//   3. ordered compaction of {key < T} and the first ties {key == T} in index order into LDS (key 32 bit, index 16 bit);
//   4. stable LSD radix sort of those <= 12 288 pairs in LDS, 4 passes of 8 bits: every wave owns a contiguous run, ranks
//      its keys per digit with the same ballot match (no atomics), digit-major / wave-minor scan of the 16 x 256 counts,
//      scatter to the other LDS buffer; a pass whose keys all share one digit is skipped;
//   5. the indices (and optionally the scores) go out in order.
// Same result as a stable descending sort cut at topn (ties in ascending index order). Can run when n <= 40 960 and
// min(topn, n) <= 12 288; WHERE it runs is decided by measurement (topk_preferred below).
constexpr int TK_THREADS = 1024, TK_WAVES = 16, TK_CAP = 12288, TK_MAXR = 40, TK_SEGR = TK_CAP / (TK_WAVES * 64);
// (the kernel is instantiated for <= 24 and <= 40 register-resident keys per lane: rows of <= 24 576 / <= 40 960 scores)
struct TopkSmem {
  unsigned keys[2][TK_CAP];
  unsigned short idx[2][TK_CAP];
  unsigned short cnt[TK_WAVES][256];
  unsigned hist[256];
  unsigned tot[256];
  unsigned wsum[2][2][TK_WAVES];
  unsigned bc[4];
};

// lanes of `act` whose 8-bit digit equals this lane's
__device__ __forceinline__ unsigned long long match_digit(unsigned d, bool in, unsigned long long act) {
  unsigned long long m = act;
#pragma unroll
  for (int bit = 0; bit < 8; ++bit) {
    const bool one = (d >> bit) & 1u;
    const unsigned long long bal = __ballot(in && one);
    m &= one ? bal : ~bal;
  }
  return m;
}

template <int MAXR>
__global__ void __launch_bounds__(TK_THREADS)
topk_sort_kernel(const float* __restrict__ scores, int n, int topn, int* __restrict__ order, int order_stride,
                 float* __restrict__ sorted_scores) {
  extern __shared__ __attribute__((aligned(16))) unsigned char tk_raw[];
  TopkSmem& sm = *reinterpret_cast<TopkSmem*>(tk_raw);
  const int b = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  const float* sc = scores + (long)b * n;
  const int m = topn < n ? topn : n;  // pairs that get sorted
  const int R = (n + TK_THREADS - 1) / TK_THREADS;

  if (n > topn) {
    unsigned K[MAXR];
#pragma unroll
    for (int j = 0; j < MAXR; ++j) {
      const int i = j * TK_THREADS + tid;
      K[j] = (j < R && i < n) ? ~monotone_bits(sc[i]) : 0xFFFFFFFFu;
    }
    // ---- radix select: T = key of rank topn (ascending), k = how many keys == T belong to the first topn ----
    unsigned prefix = 0;
    int k = topn;
    for (int shift = 24; shift >= 0; shift -= 8) {
      if (tid < 256) sm.hist[tid] = 0;
      __syncthreads();
#pragma unroll
      for (int j = 0; j < MAXR; ++j) {
        if (j < R) {
          const int i = j * TK_THREADS + tid;
          const bool in = i < n && (shift == 24 || (K[j] >> (shift + 8)) == prefix);
          const unsigned d = (K[j] >> shift) & 255u;
          const unsigned long long act = __ballot(in);
          if (act) {
            const unsigned long long mm = match_digit(d, in, act);
            if (in && (mm & lt_mask) == 0) atomicAdd(&sm.hist[d], (unsigned)__builtin_popcountll(mm));
          }
        }
      }
      __syncthreads();
      if (wave == 0) {
        unsigned c[4], s = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          c[q] = sm.hist[4 * lane + q];
          s += c[q];
        }
        unsigned incl = s;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
          const unsigned t = __shfl_up(incl, off);
          if (lane >= off) incl += t;
        }
        unsigned before = incl - s;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if ((int)before < k && k <= (int)(before + c[q])) {
            sm.bc[0] = 4 * lane + q;
            sm.bc[1] = (unsigned)k - before;
          }
          before += c[q];
        }
      }
      __syncthreads();
      prefix = (prefix << 8) | sm.bc[0];
      k = (int)sm.bc[1];
      __syncthreads();
    }
    const unsigned T = prefix;
    // ---- ordered compaction (index order): keys < T, and the first k keys == T ----
    unsigned run_eq = 0, run_sel = 0;
#pragma unroll
    for (int j = 0; j < MAXR; ++j) {
      if (j < R) {
      const int i = j * TK_THREADS + tid;
      const bool valid = i < n;
      const bool lt = valid && K[j] < T, eq = valid && K[j] == T;
      const unsigned long long be = __ballot(eq);
      unsigned(*ws)[TK_WAVES] = sm.wsum[j & 1];
      if (lane == 0) ws[0][wave] = (unsigned)__builtin_popcountll(be);
      __syncthreads();
      unsigned eq_before = run_eq, eq_all = 0;
#pragma unroll
      for (int w = 0; w < TK_WAVES; ++w) {
        const unsigned cw = ws[0][w];
        if (w < wave) eq_before += cw;
        eq_all += cw;
      }
      eq_before += (unsigned)__builtin_popcountll(be & lt_mask);
      const bool sel = lt || (eq && (int)eq_before < k);
      const unsigned long long bs = __ballot(sel);
      if (lane == 0) ws[1][wave] = (unsigned)__builtin_popcountll(bs);
      __syncthreads();
      unsigned pos = run_sel, sel_all = 0;
#pragma unroll
      for (int w = 0; w < TK_WAVES; ++w) {
        const unsigned cw = ws[1][w];
        if (w < wave) pos += cw;
        sel_all += cw;
      }
      pos += (unsigned)__builtin_popcountll(bs & lt_mask);
      if (sel) {
        sm.keys[0][pos] = K[j];
        sm.idx[0][pos] = (unsigned short)i;
      }
      run_eq += eq_all;
      run_sel += sel_all;
      }
    }
  } else {
    for (int i = tid; i < n; i += TK_THREADS) {
      sm.keys[0][i] = ~monotone_bits(sc[i]);
      sm.idx[0][i] = (unsigned short)i;
    }
  }
  __syncthreads();

  // ---- stable LSD radix sort of m (key, index) pairs in LDS ----
  const int seg = (m + TK_WAVES * 64 - 1) / (TK_WAVES * 64) * 64;  // elements per wave, a multiple of 64
  const int RW = seg / 64;
  int cur = 0;
  for (int shift = 0; shift < 32; shift += 8) {
    unsigned kk[TK_SEGR];
    unsigned short ii[TK_SEGR], rk[TK_SEGR];
#pragma unroll
    for (int q = 0; q < 4; ++q) sm.cnt[wave][4 * lane + q] = 0;
#pragma unroll
    for (int r = 0; r < TK_SEGR; ++r) {
      const int e = wave * seg + r * 64 + lane;
      const bool valid = r < RW && e < m;
      kk[r] = valid ? sm.keys[cur][e] : 0u;
      ii[r] = valid ? sm.idx[cur][e] : (unsigned short)0;
      const unsigned d = (kk[r] >> shift) & 255u;
      const unsigned long long act = __ballot(valid);
      unsigned base = 0;
      unsigned long long mm = 1ull << lane;
      if (act) {
        mm = match_digit(d, valid, act);
        if (valid && (mm & lt_mask) == 0) {
          base = sm.cnt[wave][d];
          sm.cnt[wave][d] = (unsigned short)(base + (unsigned)__builtin_popcountll(mm));
        }
      }
      const int leader = valid ? __builtin_ctzll(mm) : lane;
      base = __shfl(base, leader);
      rk[r] = (unsigned short)(base + (unsigned)__builtin_popcountll(mm & lt_mask));
    }
    __syncthreads();
    if (tid < 256) {  // digit-major, wave-minor: exclusive prefix over the waves, total per digit
      unsigned run = 0;
#pragma unroll
      for (int w = 0; w < TK_WAVES; ++w) {
        const unsigned c = sm.cnt[w][tid];
        sm.cnt[w][tid] = (unsigned short)run;
        run += c;
      }
      sm.tot[tid] = run;
    }
    __syncthreads();
    if (wave == 0) {
      unsigned c[4], s = 0;
      bool one_digit = false;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        c[q] = sm.tot[4 * lane + q];
        s += c[q];
        one_digit |= c[q] == (unsigned)m;
      }
      unsigned incl = s;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const unsigned t = __shfl_up(incl, off);
        if (lane >= off) incl += t;
      }
      unsigned before = incl - s;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        sm.tot[4 * lane + q] = before;
        before += c[q];
      }
      const unsigned long long any = __ballot(one_digit);
      if (lane == 0) sm.bc[2] = any ? 1u : 0u;
    }
    __syncthreads();
    const bool skip = sm.bc[2] != 0;  // every key has the same digit: the pass is the identity
    if (!skip) {
#pragma unroll
      for (int r = 0; r < TK_SEGR; ++r) {
        const int e = wave * seg + r * 64 + lane;
        if (r < RW && e < m) {
          const unsigned d = (kk[r] >> shift) & 255u;
          const unsigned dst = sm.tot[d] + sm.cnt[wave][d] + rk[r];
          sm.keys[cur ^ 1][dst] = kk[r];
          sm.idx[cur ^ 1][dst] = ii[r];
        }
      }
      cur ^= 1;
    }
    __syncthreads();
  }
  for (int e = tid; e < m; e += TK_THREADS) {
    order[(long)b * order_stride + e] = (int)sm.idx[cur][e];
    if (sorted_scores) sorted_scores[(long)b * order_stride + e] = from_monotone_bits(~sm.keys[cur][e]);
  }
}

// ---- multi-workgroup top-k + sort for long rows (round 5): a sample sort over the whole chip --------------------------------
// The single-workgroup kernel above is four SIMDs: 122 us on the proposal layer's rows (4 x 21 546 -> 12 000), where a
// device-wide library radix sort (rocPRIM, rounds 1-4: 8-10 launches) took 88 us. This one spreads ONE row over many CUs in
// four short launches (45 us) and is the only sort of rows longer than 4 096 scores -- no library is linked:
//   composite key  c = (~monotone(score)) << 32 | index   -- unique, and ascending c = descending score, ties in ascending
//                                                            index order (the stable descending sort's order)
//   1. ss_sample_kernel    (B workgroups)       1 024 (2 048) evenly spaced keys of the row, bitonic-sorted in registers ->
//                                               every 8th is a splitter: NB = 128 (256 for rows of more than 32 768 scores) buckets of
//                                               ~n / NB keys, +-25 % (unique keys: ties cannot pile up in one bucket; the
//                                               bucket sort below is quadratic in the bucket size, so balance is speed)
//   2. ss_classify_kernel  (G x B workgroups)   1 024 keys per workgroup (more for rows beyond 65 536 scores, so that the G x NB
//                                               counters below fit LDS): bucket by binary search over the splitters (LDS),
//                                               per-workgroup bucket histogram -> global
//   3. ss_scatter_kernel   (G x B)              bucket bases from the histograms (every workgroup sums them itself: G x 128
//                                               counters), keys scattered to their bucket's range; buckets that start at or
//                                               behind rank topn are dropped
//   4. ss_bucket_kernel    (128 x B)            each surviving bucket ranked by counting in LDS (unique keys -> a total order:
//                                               the scatter's atomic order inside a bucket does not matter) and written out:
//                                               order[start + rank], optionally the score
// Buckets larger than SS_CAP keys (rows of more than ~400 000 scores, which this model never produces: its longest row is
// 50 400) are ranked straight from global memory -- correct, slow.
constexpr int SS_NBMAX = 256, SS_CHUNK = 1024, SS_CAP = 2048, SS_HMAX = 16384;  // (G * NB <= SS_HMAX histogram counters)
__host__ __device__ inline int ss_nb(int n) { return n > 32768 ? 256 : 128; }
// keys per classify / scatter workgroup: SS_CHUNK, or the multiple of it that keeps G * NB <= SS_HMAX
inline int ss_chunk(int n) {
  const int gmax = SS_HMAX / ss_nb(n);
  return SS_CHUNK * (int)(((long)n + (long)SS_CHUNK * gmax - 1) / ((long)SS_CHUNK * gmax));
}
__device__ __forceinline__ unsigned long long ss_key(float score, int index) {
  return ((unsigned long long)(~monotone_bits(score)) << 32) | (unsigned)index;
}

// how many of the n8 (a multiple of 8; padded with ~0) keys at `k` are smaller than c: eight broadcast keys per trip, the four
// 16-byte LDS reads of a trip independent of each other (one read per trip behind its own wait was 20 us per launch)
__device__ __forceinline__ unsigned ss_count_less(const unsigned long long* k, unsigned n8, unsigned long long c) {
  unsigned cnt = 0;
  for (unsigned j = 0; j < n8; j += 8) {
    const ulonglong2 a = *(const ulonglong2*)&k[j], b = *(const ulonglong2*)&k[j + 2];
    const ulonglong2 d = *(const ulonglong2*)&k[j + 4], e = *(const ulonglong2*)&k[j + 6];
    cnt += (a.x < c ? 1u : 0u) + (a.y < c ? 1u : 0u) + (b.x < c ? 1u : 0u) + (b.y < c ? 1u : 0u) + (d.x < c ? 1u : 0u) +
           (d.y < c ? 1u : 0u) + (e.x < c ? 1u : 0u) + (e.y < c ? 1u : 0u);
  }
  return cnt;
}

template <int S_>
__global__ void __launch_bounds__(S_ / 2)
ss_sample_kernel(const float* __restrict__ scores, int n, int NB, unsigned long long* __restrict__ splitters) {
  __shared__ __attribute__((aligned(16))) unsigned long long smp[S_];
  const int b = blockIdx.x, t = threadIdx.x, e0 = 2 * t;
  // bitonic sort, ascending, with the keys in REGISTERS: lane t holds elements 2t and 2t + 1. A compare-exchange at distance
  // j = 1 is between the lane's own two keys, at 2 <= j <= 64 between this lane and lane t ^ (j / 2) of the same wave (a
  // shuffle per key), and only the steps with j >= 128 (10 of 66 at S = 2 048) cross waves and go through LDS (one 16-byte
  // write and one 16-byte read per lane). The launch is VALU-bound on its ONE CU -- 66 steps x 16 waves x ~30 instructions --
  // so the step's form (this, or every step through LDS) moved it by < 1 us and the sample count is what sets its time.
  const int ia = (int)((long)e0 * n / S_), ib = (int)((long)(e0 + 1) * n / S_);
  unsigned long long x = ss_key(scores[(long)b * n + ia], ia), y = ss_key(scores[(long)b * n + ib], ib);
  for (int k = 2; k <= S_; k <<= 1) {
    const bool up = (e0 & k) == 0;  // (bit k of 2t + 1 is the same: k >= 2)
    for (int j = k >> 1; j > 0; j >>= 1) {
      if (j == 1) {
        const unsigned long long lo = x < y ? x : y, hi = x < y ? y : x;
        x = up ? lo : hi;
        y = up ? hi : lo;
        continue;
      }
      unsigned long long px, py;
      if (j <= 64) {
        px = __shfl_xor(x, j >> 1);
        py = __shfl_xor(y, j >> 1);
      } else {
        *(ulonglong2*)&smp[e0] = make_ulonglong2(x, y);
        __syncthreads();
        const ulonglong2 pr = *(const ulonglong2*)&smp[e0 ^ j];
        px = pr.x;
        py = pr.y;
        __syncthreads();
      }
      const bool keep_min = ((e0 & j) == 0) == up;
      x = (px < x) == keep_min ? px : x;
      y = (py < y) == keep_min ? py : y;
    }
  }
  *(ulonglong2*)&smp[e0] = make_ulonglong2(x, y);
  __syncthreads();
  if (t < NB - 1) splitters[(long)b * SS_NBMAX + t] = smp[(t + 1) * (S_ / NB)];
}

__device__ __forceinline__ int ss_bucket(const unsigned long long* sp, int NB, unsigned long long c) {
  // number of splitters <= c (sp[NB - 1 ..] hold ~0)
  int lo = 0;
#pragma unroll
  for (int step = SS_NBMAX / 2; step > 0; step >>= 1)
    if (lo + step <= NB - 1 && sp[lo + step - 1] <= c) lo += step;
  return lo;
}

__global__ void __launch_bounds__(256)
ss_classify_kernel(const float* __restrict__ scores, int n, int NB, int chunk, const unsigned long long* __restrict__ splitters,
                   unsigned* __restrict__ hist) {
  __shared__ unsigned long long sp[SS_NBMAX];
  __shared__ unsigned cnt[SS_NBMAX];
  const int b = blockIdx.y, g = blockIdx.x, t = threadIdx.x;
  sp[t] = t < NB - 1 ? splitters[(long)b * SS_NBMAX + t] : ~0ull;
  cnt[t] = 0;
  __syncthreads();
  for (int k = 0; k < chunk / 256; ++k) {
    const int i = g * chunk + k * 256 + t;
    if (i < n) atomicAdd(&cnt[ss_bucket(sp, NB, ss_key(scores[(long)b * n + i], i))], 1u);
  }
  __syncthreads();
  if (t < NB) hist[((long)b * gridDim.x + g) * NB + t] = cnt[t];
}

__global__ void __launch_bounds__(256)
ss_scatter_kernel(const float* __restrict__ scores, int n, int topn, int NB, int chunk, const unsigned long long* __restrict__ splitters,
                  const unsigned* __restrict__ hist, unsigned* __restrict__ bucket_start,
                  unsigned long long* __restrict__ bucketed) {
  __shared__ unsigned long long sp[SS_NBMAX];
  __shared__ unsigned base[SS_NBMAX], tot[SS_NBMAX], scan[SS_NBMAX], hs[SS_HMAX];
  const int b = blockIdx.y, g = blockIdx.x, G = gridDim.x, t = threadIdx.x;
  // every workgroup sums the G x NB counters itself: one coalesced sweep into LDS (independent loads: one round trip)
  for (int i = t; i < G * NB; i += 256) hs[i] = hist[(long)b * G * NB + i];
  sp[t] = t < NB - 1 ? splitters[(long)b * SS_NBMAX + t] : ~0ull;
  __syncthreads();
  {
    unsigned before = 0, total = 0;
    if (t < NB)
      for (int gg = 0; gg < G; ++gg) {
        const unsigned h = hs[gg * NB + t];
        before += gg < g ? h : 0u;
        total += h;
      }
    base[t] = before;
    tot[t] = total;
  }
  __syncthreads();
  if (t < 64) {  // exclusive scan over the (up to 256) bucket totals: four per lane, one wave
    const unsigned a0 = tot[4 * t], a1 = tot[4 * t + 1], a2 = tot[4 * t + 2], a3 = tot[4 * t + 3];
    const unsigned sum = a0 + a1 + a2 + a3;
    unsigned incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned v = __shfl_up(incl, o);
      if (t >= o) incl += v;
    }
    const unsigned excl = incl - sum;
    scan[4 * t] = excl;
    scan[4 * t + 1] = excl + a0;
    scan[4 * t + 2] = excl + a0 + a1;
    scan[4 * t + 3] = excl + a0 + a1 + a2;
  }
  __syncthreads();
  base[t] += scan[t];
  if (g == 0 && t < NB) {
    bucket_start[((long)b * SS_NBMAX + t) * 2] = scan[t];
    bucket_start[((long)b * SS_NBMAX + t) * 2 + 1] = tot[t];
  }
  __syncthreads();
  for (int k = 0; k < chunk / 256; ++k) {
    const int i = g * chunk + k * 256 + t;
    if (i < n) {
      const unsigned long long c = ss_key(scores[(long)b * n + i], i);
      const int bk = ss_bucket(sp, NB, c);
      if (scan[bk] < (unsigned)topn) bucketed[(long)b * n + atomicAdd(&base[bk], 1u)] = c;
    }
  }
}

__global__ void __launch_bounds__(256)
ss_bucket_kernel(const unsigned long long* __restrict__ bucketed, int n, int topn, const unsigned* __restrict__ bucket_start,
                 int* __restrict__ order, int order_stride, float* __restrict__ sorted_scores) {
  __shared__ __attribute__((aligned(16))) unsigned long long keys[SS_CAP + 8];
  const int b = blockIdx.y, bk = blockIdx.x, t = threadIdx.x;
  const unsigned start = bucket_start[((long)b * SS_NBMAX + bk) * 2], size = bucket_start[((long)b * SS_NBMAX + bk) * 2 + 1];
  if (start >= (unsigned)topn || size == 0) return;
  const unsigned long long* src = bucketed + (long)b * n + start;
  const bool in_lds = size <= SS_CAP;
  const unsigned n8 = (size + 7) & ~7u;
  if (in_lds) {
    for (unsigned i = t; i < n8; i += 256) keys[i] = i < size ? src[i] : ~0ull;
    __syncthreads();
  }
  for (unsigned i = t; i < size; i += 256) {
    const unsigned long long c = in_lds ? keys[i] : src[i];
    unsigned rank = 0;
    if (in_lds) {
      rank = ss_count_less(keys, n8, c);
    } else {
      for (unsigned j = 0; j < size; ++j) rank += src[j] < c ? 1u : 0u;
    }
    const unsigned pos = start + rank;
    if (pos < (unsigned)topn) {
      order[(long)b * order_stride + pos] = (int)(unsigned)c;
      if (sorted_scores) sorted_scores[(long)b * order_stride + pos] = from_monotone_bits(~(unsigned)(c >> 32));
    }
  }
}

struct SsPlan {
  size_t splitters, hist, bucket_start, bucketed, total;
  int G, chunk;
};
SsPlan ss_plan(int B, int n) {
  SsPlan p;
  p.chunk = ss_chunk(n);
  p.G = (n + p.chunk - 1) / p.chunk;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    const size_t o = off;
    off += dana_align_up(bytes, 256);
    return o;
  };
  p.splitters = take((size_t)B * SS_NBMAX * 8);
  p.hist = take((size_t)B * p.G * SS_NBMAX * 4);
  p.bucket_start = take((size_t)B * SS_NBMAX * 2 * 4);
  p.bucketed = take((size_t)B * n * 8);
  p.total = off;
  return p;
}

int ss_launch(const float* scores, int B, int n, int topn, int* order, int order_stride, float* sorted_scores, void* workspace,
              hipStream_t s) {
  const SsPlan p = ss_plan(B, n);
  char* ws = (char*)workspace;
  unsigned long long* splitters = (unsigned long long*)(ws + p.splitters);
  unsigned* hist = (unsigned*)(ws + p.hist);
  unsigned* bstart = (unsigned*)(ws + p.bucket_start);
  unsigned long long* bucketed = (unsigned long long*)(ws + p.bucketed);
  const int m = topn < n ? topn : n, NB = ss_nb(n);
  if (NB == 128)  // (measured: 12.5 vs 17 us for the launch; the 128 buckets stay balanced enough at 8 samples per splitter)
    ss_sample_kernel<1024><<<B, 512, 0, s>>>(scores, n, NB, splitters);
  else
    ss_sample_kernel<2048><<<B, 1024, 0, s>>>(scores, n, NB, splitters);
  ss_classify_kernel<<<dim3(p.G, B), 256, 0, s>>>(scores, n, NB, p.chunk, splitters, hist);
  ss_scatter_kernel<<<dim3(p.G, B), 256, 0, s>>>(scores, n, m, NB, p.chunk, splitters, hist, bstart, bucketed);
  ss_bucket_kernel<<<dim3(NB, B), 256, 0, s>>>(bucketed, n, m, bstart, order, order_stride, sorted_scores);
  return 0;
}

// Measured (tools/topk_bench.py, one MI355X): ONE workgroup per row is all of 4 SIMDs -- 19 us for a row of 300 scores
// (detection post-processing; the four launches of the sample sort are 30 us there), but 122 us for 4 x 21 546 -> 12 000 and
// 177 us for 2 x 37 800 -> 12 000 against the sample sort's 45 us: the ballot matches of ~100 000 (key, pass) pairs serialise
// on one CU. So the single-workgroup kernel takes rows of <= 4 096 scores and the sample sort every longer row.
bool topk_preferred(int n, int topn);
bool topk_supported(int n, int topn) {
  const int m = topn < n ? topn : n;
  return n > 0 && n <= TK_MAXR * TK_THREADS && n < 65536 && m <= TK_CAP;
}

bool topk_preferred(int n, int topn) {
  if (!topk_supported(n, topn) || g_sort_mode == 1) return false;
  return g_sort_mode == 2 || n <= 4096;
}

int topk_launch(const float* scores, int B, int n, int topn, int* order, int order_stride, float* sorted_scores,
                hipStream_t s) {
  static DeviceOnce attr;  // (per device: common.h)
  attr.once([&] {
    const hipError_t e0 = hipFuncSetAttribute((const void*)topk_sort_kernel<24>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(TopkSmem));
    const hipError_t e1 = hipFuncSetAttribute((const void*)topk_sort_kernel<TK_MAXR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(TopkSmem));
    return e0 != hipSuccess ? e0 : e1;
  });
  if (n <= 24 * TK_THREADS)
    topk_sort_kernel<24><<<B, TK_THREADS, sizeof(TopkSmem), s>>>(scores, n, topn, order, order_stride, sorted_scores);
  else
    topk_sort_kernel<TK_MAXR><<<B, TK_THREADS, sizeof(TopkSmem), s>>>(scores, n, topn, order, order_stride, sorted_scores);
  return 0;
}

}  // namespace

extern "C" {

int dana_rpn_decode(const float* cls, long cls_sb, long cls_sc, long cls_sp, int cls_is_prob, const float* bbox,
                    long bbox_sb, long bbox_sc, long bbox_sp, const float* im_info, const float* base_anchors,
                    int B, int A, int H, int W, int feat_stride, float* proposals, float* scores,
                    dana_stream_t stream) {
  DANA_CHECK_ARG(B >= 0 && A > 0 && H > 0 && W > 0, "dana_rpn_decode: bad shape B=%d A=%d H=%d W=%d", B, A, H, W);
  if (B == 0) return DANA_OK;
  DANA_CHECK_ARG(cls && bbox && im_info && base_anchors && proposals && scores, "dana_rpn_decode: null pointer");
  const int n = H * W * A;
  dim3 grid(dana_ceil_div(n, 256), B);
  rpn_decode_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(cls, cls_sb, cls_sc, cls_sp, cls_is_prob, bbox, bbox_sb,
                                                           bbox_sc, bbox_sp, im_info, base_anchors, A, H, W,
                                                           feat_stride, (float4*)proposals, scores);
  DANA_CHECK_LAUNCH("dana_rpn_decode");
  return DANA_OK;
}

size_t dana_sort_desc_workspace_bytes(int B, int n) { return dana_topk_desc_workspace_bytes(B, n, n); }

// Stable descending sort of each row of scores[B][n]; order[B][n] = source index within the row.
int dana_sort_desc(const float* scores, int B, int n, int* order, float* sorted_scores, void* workspace,
                   size_t workspace_bytes, dana_stream_t stream) {
  DANA_CHECK_ARG(B >= 0 && n >= 0, "dana_sort_desc: bad shape");
  return dana_topk_desc(scores, B, n, n, order, n, sorted_scores, workspace, workspace_bytes, stream);
}

int dana_set_sort_mode(int mode) {
  DANA_CHECK_ARG(mode >= 0 && mode <= 2, "dana_set_sort_mode: mode %d", mode);
  g_sort_mode = mode;
  return DANA_OK;
}

int dana_topk_desc(const float* scores, int B, int n, int topn, int* order, int order_stride, float* sorted_scores,
                   void* workspace, size_t workspace_bytes, dana_stream_t stream) {
  DANA_CHECK_ARG(B >= 0 && n >= 0 && topn >= 0 && order_stride >= (topn < n ? topn : n), "dana_topk_desc: bad shape");
  if (B == 0 || n == 0 || topn == 0) return DANA_OK;
  DANA_CHECK_ARG(scores && order, "dana_topk_desc: null pointer");
  if (topk_preferred(n, topn)) {  // short rows: the single-workgroup select + LDS sort, one launch
    topk_launch(scores, B, n, topn, order, order_stride, sorted_scores, (hipStream_t)stream);
    DANA_CHECK_LAUNCH("dana_topk_desc");
    return DANA_OK;
  }
  const size_t need = ss_plan(B, n).total;
  if (!workspace || workspace_bytes < need) {
    dana_set_error("dana_topk_desc: workspace %zu < %zu", workspace_bytes, need);
    return DANA_ERR_WORKSPACE;
  }
  // long rows: the multi-workgroup sample sort, four launches
  ss_launch(scores, B, n, topn, order, order_stride, sorted_scores, workspace, (hipStream_t)stream);
  DANA_CHECK_LAUNCH("dana_topk_desc(sample sort)");
  return DANA_OK;
}

size_t dana_topk_desc_workspace_bytes(int B, int n, int topn) {
  if (B <= 0 || n <= 0 || topn <= 0) return 0;
  return ss_plan(B, n).total;  // (also for rows the single-workgroup kernel takes: dana_set_sort_mode may switch)
}

int dana_gather_boxes(const float* src, const int* order, int B, int n, int order_stride, int topn, float* dst,
                      dana_stream_t stream) {
  DANA_CHECK_ARG(B >= 0 && n >= 0 && topn >= 0 && topn <= n, "dana_gather_boxes: bad shape");
  if (B == 0 || topn == 0) return DANA_OK;
  DANA_CHECK_ARG(src && order && dst, "dana_gather_boxes: null pointer");
  dim3 grid(dana_ceil_div(topn, 256), B);
  gather_boxes_kernel<<<grid, 256, 0, (hipStream_t)stream>>>((const float4*)src, order, n, order_stride, topn,
                                                             (float4*)dst);
  DANA_CHECK_LAUNCH("dana_gather_boxes");
  return DANA_OK;
}

int dana_rois_assemble(const float* sorted_boxes, const int* keep, const int* num_keep, int B, int topn,
                       int keep_stride, int post_n, float* rois, dana_stream_t stream) {
  DANA_CHECK_ARG(B >= 0 && topn >= 0 && post_n >= 0, "dana_rois_assemble: bad shape");
  if (B == 0 || post_n == 0) return DANA_OK;
  DANA_CHECK_ARG(sorted_boxes && keep && num_keep && rois, "dana_rois_assemble: null pointer");
  dim3 grid(dana_ceil_div(post_n, 256), B);
  rois_assemble_kernel<<<grid, 256, 0, (hipStream_t)stream>>>((const float4*)sorted_boxes, keep, num_keep, topn,
                                                              keep_stride, post_n, rois);
  DANA_CHECK_LAUNCH("dana_rois_assemble");
  return DANA_OK;
}

// ---- whole proposal layer in one call --------------------------------------------------------
struct ProposalPlan {
  size_t proposals, scores, order, sorted_boxes, keep, num_keep, sort_ws, nms_ws, total;
  int topn;
};
static ProposalPlan proposal_plan(int B, int n, int pre_nms_topn, int post_nms_topn) {
  ProposalPlan p;
  // proposal_layer.py:148: `pre_nms_topN < scores_keep.numel()` compares against the WHOLE batch's count
  p.topn = (pre_nms_topn > 0 && (long)pre_nms_topn < (long)B * n) ? (pre_nms_topn < n ? pre_nms_topn : n) : n;
  size_t o = 0;
  auto take = [&](size_t bytes) {
    size_t at = o;
    o += dana_align_up(bytes, 256);
    return at;
  };
  p.proposals = take((size_t)B * n * 16);
  p.scores = take((size_t)B * n * 4);
  p.order = take((size_t)B * n * 4);
  p.sorted_boxes = take((size_t)B * p.topn * 16);
  int mk = (post_nms_topn > 0 && post_nms_topn < p.topn) ? post_nms_topn : p.topn;
  p.keep = take((size_t)B * mk * 4);
  p.num_keep = take((size_t)B * 4);
  p.sort_ws = take(dana_topk_desc_workspace_bytes(B, n, p.topn));
  p.nms_ws = take(dana_nms_workspace_bytes(p.topn, B));
  p.total = o;
  return p;
}

size_t dana_proposal_layer_workspace_bytes(int B, int A, int H, int W, int pre_nms_topn, int post_nms_topn) {
  if (B <= 0 || A <= 0 || H <= 0 || W <= 0) return 0;
  return proposal_plan(B, H * W * A, pre_nms_topn, post_nms_topn).total;
}

int dana_proposal_layer(const float* cls, long cls_sb, long cls_sc, long cls_sp, int cls_is_prob, const float* bbox,
                        long bbox_sb, long bbox_sc, long bbox_sp, const float* im_info, const float* base_anchors,
                        int B, int A, int H, int W, int feat_stride, int pre_nms_topn, int post_nms_topn,
                        float nms_thresh, int nms_inclusive, float* rois, void* workspace, size_t workspace_bytes,
                        dana_stream_t stream) {
  DANA_CHECK_ARG(B >= 0 && A > 0 && H > 0 && W > 0 && post_nms_topn > 0, "dana_proposal_layer: bad shape");
  if (B == 0) return DANA_OK;
  const int n = H * W * A;
  ProposalPlan p = proposal_plan(B, n, pre_nms_topn, post_nms_topn);
  if (!workspace || workspace_bytes < p.total) {
    dana_set_error("dana_proposal_layer: workspace %zu < %zu", workspace_bytes, p.total);
    return DANA_ERR_WORKSPACE;
  }
  char* ws = (char*)workspace;
  float* proposals = (float*)(ws + p.proposals);
  float* scores = (float*)(ws + p.scores);
  int* order = (int*)(ws + p.order);
  float* sorted_boxes = (float*)(ws + p.sorted_boxes);
  int* keep = (int*)(ws + p.keep);
  int* num_keep = (int*)(ws + p.num_keep);
  const int mk = post_nms_topn < p.topn ? post_nms_topn : p.topn;
  int rc = dana_rpn_decode(cls, cls_sb, cls_sc, cls_sp, cls_is_prob, bbox, bbox_sb, bbox_sc, bbox_sp, im_info,
                           base_anchors, B, A, H, W, feat_stride, proposals, scores, stream);
  if (rc) return rc;
  // the first pre_nms_topN of every image's descending score order (proposal_layer.py:135-150): the single-workgroup
  // select + sort for short rows, the multi-workgroup sample sort for the real ones (dana_topk_desc)
  rc = dana_topk_desc(scores, B, n, p.topn, order, n, nullptr, ws + p.sort_ws, p.nms_ws - p.sort_ws, stream);
  if (rc) return rc;
  rc = dana_gather_boxes(proposals, order, B, n, n, p.topn, sorted_boxes, stream);
  if (rc) return rc;
  rc = dana_nms(sorted_boxes, p.topn, B, nms_thresh, nms_inclusive, mk, keep, mk, num_keep, ws + p.nms_ws,
                p.total - p.nms_ws, stream);
  if (rc) return rc;
  return dana_rois_assemble(sorted_boxes, keep, num_keep, B, p.topn, mk, post_nms_topn, rois, stream);
}

// ---- inference post-processing: inference.py:106-140 + utils.py:312-317 -------------
struct DetectPlan {
  size_t boxes, scores, order, sorted_scores, sorted_boxes, keep, meta, sort_ws, nms_ws, total;
};
static DetectPlan detect_plan(int R) {
  DetectPlan p;
  size_t o = 0;
  auto take = [&](size_t bytes) {
    size_t at = o;
    o += dana_align_up(bytes, 256);
    return at;
  };
  p.boxes = take((size_t)R * 16);
  p.scores = take((size_t)R * 4);
  p.order = take((size_t)R * 4);
  p.sorted_scores = take((size_t)R * 4);
  p.sorted_boxes = take((size_t)R * 16);
  p.keep = take((size_t)R * 4);
  p.meta = take(16);
  p.sort_ws = take(dana_sort_desc_workspace_bytes(1, R));
  p.nms_ws = take(dana_nms_workspace_bytes(R, 1));
  p.total = o;
  return p;
}

size_t dana_detect_postprocess_workspace_bytes(int R) { return R > 0 ? detect_plan(R).total : 0; }

// dets[R][5] = (x1, y1, x2, y2, score) of the kept detections in descending score order; meta[0] = number of
// rows passing the threshold, meta[1] = number of rows NMS kept among ALL R sorted rows: the caller keeps the
// first `kept positions < meta[0]` (rows below the threshold sort last and can never suppress a valid row).
int dana_detect_postprocess(const float* rois, const float* cls_prob, const float* bbox_pred, const float* im_info,
                            int R, const float* stds4, const float* means4, int normalize, float score_thresh,
                            float nms_thresh, int nms_inclusive, float* dets, int* keep_pos, int* meta,
                            void* workspace, size_t workspace_bytes, dana_stream_t stream) {
  DANA_CHECK_ARG(R >= 0, "dana_detect_postprocess: bad R");
  DANA_CHECK_ARG(meta, "dana_detect_postprocess: null meta");
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(meta, 0, 2 * sizeof(int), s) != hipSuccess) {
    dana_set_error("dana_detect_postprocess: memset failed");
    return DANA_ERR_HIP;
  }
  if (R == 0) return DANA_OK;
  DANA_CHECK_ARG(rois && cls_prob && bbox_pred && im_info && stds4 && means4 && dets && keep_pos,
                 "dana_detect_postprocess: null pointer");
  const DetectPlan p = detect_plan(R);
  if (!workspace || workspace_bytes < p.total) {
    dana_set_error("dana_detect_postprocess: workspace %zu < %zu", workspace_bytes, p.total);
    return DANA_ERR_WORKSPACE;
  }
  char* ws = (char*)workspace;
  float* boxes = (float*)(ws + p.boxes);
  float* scores = (float*)(ws + p.scores);
  int* order = (int*)(ws + p.order);
  float* sorted_scores = (float*)(ws + p.sorted_scores);
  float* sorted_boxes = (float*)(ws + p.sorted_boxes);
  const float4 sd = make_float4(stds4[0], stds4[1], stds4[2], stds4[3]);
  const float4 mn = make_float4(means4[0], means4[1], means4[2], means4[3]);
  detect_decode_kernel<<<dana_ceil_div(R, 256), 256, 0, s>>>(rois, cls_prob, bbox_pred, im_info, R, sd, mn, normalize,
                                                            score_thresh, (float4*)boxes, scores, meta);
  DANA_CHECK_LAUNCH("dana_detect_postprocess(decode)");
  int rc = dana_sort_desc(scores, 1, R, order, sorted_scores, ws + p.sort_ws, p.nms_ws - p.sort_ws, stream);
  if (rc) return rc;
  rc = dana_gather_boxes(boxes, order, 1, R, R, R, sorted_boxes, stream);
  if (rc) return rc;
  rc = dana_nms(sorted_boxes, R, 1, nms_thresh, nms_inclusive, R, keep_pos, R, meta + 1, ws + p.nms_ws,
                p.total - p.nms_ws, stream);
  if (rc) return rc;
  // dets in sorted order for every row; the caller indexes it with keep_pos
  dets_assemble_kernel<<<dana_ceil_div(R, 256), 256, 0, s>>>((const float4*)sorted_boxes, sorted_scores, R, dets);
  DANA_CHECK_LAUNCH("dana_detect_postprocess(assemble)");
  return DANA_OK;
}

}  // extern "C"
