// Counter-based device RNG for the two training target layers: the reference draws
// np.random.permutation / np.random.rand on the host after reading the fg / bg counts back
// (anchor_target_layer.py:137-156, proposal_target_layer_cascade.py:143-175) -- the last host syncs inside the
// training forward. With these kernels the counts never leave the device: Philox-4x32-10 keyed by (seed, call offset,
// image), uniform subsets without replacement by Floyd's algorithm (k sequential draws against an LDS bitmap; k <= 256),
// with-replacement picks in parallel. Same distributions as the reference's draws, a different random stream
// (opt-in: DAnARCNN.device_rng; the default keeps np.random so that parity with the reference is testable).
#include "common.h"
#include "../../include/dana_hip.h"

namespace {

struct Philox {
  unsigned k0, k1;
  __device__ Philox(unsigned long long seed) : k0((unsigned)seed), k1((unsigned)(seed >> 32)) {}
  __device__ uint4 operator()(unsigned long long offset, unsigned stream, unsigned idx) const {
    unsigned c0 = idx, c1 = stream, c2 = (unsigned)offset, c3 = (unsigned)(offset >> 32);
    unsigned a = k0, b = k1;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
      const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
      const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ a, n1 = (unsigned)p1;
      const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ b, n3 = (unsigned)p0;
      c0 = n0;
      c1 = n1;
      c2 = n2;
      c3 = n3;
      a += 0x9E3779B9u;
      b += 0xBB67AE85u;
    }
    return make_uint4(c0, c1, c2, c3);
  }
};

__device__ __forceinline__ float u01(unsigned x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }  // [0, 1)
// uniform integer in [0, n): floor(u * n) like the reference's np.floor(np.random.rand() * n), clamped
__device__ __forceinline__ int below(unsigned x, int n) {
  const int v = (int)(u01(x) * (float)n);
  return v < n ? v : n - 1;
}

// Floyd: a uniform k-subset of [0, n) as set bits of `bits` (cleared by the caller); one thread
__device__ void floyd_subset(unsigned* bits, int n, int k, const Philox& rng, unsigned long long offset, unsigned stream) {
  for (int j = n - k; j < n; ++j) {
    const uint4 r = rng(offset, stream, (unsigned)j);
    // exact uniform integer in [0, j] from 32 random bits (multiply-shift; bias < 2^-32 * j)
    const int t = (int)(((unsigned long long)r.x * (unsigned long long)(j + 1)) >> 32);
    const bool taken = (bits[t >> 5] >> (t & 31)) & 1u;
    const int s = taken ? j : t;
    bits[s >> 5] |= 1u << (s & 31);
  }
}

__global__ void counter_add_kernel(unsigned long long* c, unsigned long long inc) {
  if (threadIdx.x == 0 && blockIdx.x == 0) c[0] += inc;
}

// grid = B, block = 256, LDS bitmap of `cand` bits. picks[b][R]; fg_taken[b]
__global__ void __launch_bounds__(256)
proposal_target_sample_kernel(const int* __restrict__ counts, int cand, int R, int fg_per, unsigned long long seed,
                              unsigned long long offset, const unsigned long long* __restrict__ offset_dev,
                              int* __restrict__ picks, int* __restrict__ fg_taken) {
  if (offset_dev) offset += offset_dev[0];  // call counter kept in device memory (hipGraph replays advance it)
  extern __shared__ unsigned bits[];  // [words] bitmap | [words] exclusive prefix of the popcounts
  const int words = (cand + 31) / 32;
  unsigned* pref = bits + words;
  const int b = blockIdx.x;
  const int nf = counts[b * 2], nb = counts[b * 2 + 1];
  const Philox rng(seed);
  int fg_n;
  if (nf > 0 && nb > 0)
    fg_n = min(fg_per, nf);
  else
    fg_n = nf > 0 ? R : 0;
  if (threadIdx.x == 0) fg_taken[b] = fg_n;
  int* out = picks + (long)b * R;
  if (nf > 0 && nb > 0) {
    // np.random.permutation(nf)[:fg_n]: a uniform subset without replacement (its order is immaterial downstream)
    for (int i = threadIdx.x; i < words; i += blockDim.x) bits[i] = 0u;
    __syncthreads();
    if (threadIdx.x == 0) {
      floyd_subset(bits, nf, fg_n, rng, offset, 4u * b);
      unsigned run = 0;
      for (int wd = 0; wd < words; ++wd) {
        pref[wd] = run;
        run += __popc(bits[wd]);
      }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nf; i += blockDim.x)  // emitted in increasing position: deterministic
      if ((bits[i >> 5] >> (i & 31)) & 1u) out[pref[i >> 5] + __popc(bits[i >> 5] & ((1u << (i & 31)) - 1u))] = i;
    // np.floor(np.random.rand(R - fg_n) * nb): with replacement
    for (int j = fg_n + threadIdx.x; j < R; j += blockDim.x) out[j] = below(rng(offset, 4u * b + 1u, (unsigned)j).x, nb);
  } else {
    const int n = nf > 0 ? nf : max(nb, 1);
    for (int j = threadIdx.x; j < R; j += blockDim.x) out[j] = below(rng(offset, 4u * b + 1u, (unsigned)j).x, n);
  }
}

// grid = B, block = 1024, LDS bitmap of `total` bits: anchor_target_layer.py:137-156 without the host
__global__ void __launch_bounds__(1024)
anchor_target_subsample_kernel(float* __restrict__ labels, const int* __restrict__ fg_list,
                               const int* __restrict__ bg_list, const int* __restrict__ counts, int B, int total,
                               int batchsize, int num_fg, unsigned long long seed, unsigned long long offset,
                               const unsigned long long* __restrict__ offset_dev,
                               float* __restrict__ inv_num_examples) {
  if (offset_dev) offset += offset_dev[0];
  extern __shared__ unsigned bits[];
  const int b = blockIdx.x;
  const int nf = counts[b * 2], nb = counts[b * 2 + 1];
  const Philox rng(seed);
  const int words = (total + 31) / 32;
  float* lab = labels + (long)b * total;
  if (nf > num_fg) {  // keep a uniform num_fg-subset of the positives, disable the rest (:137-141)
    for (int i = threadIdx.x; i < words; i += blockDim.x) bits[i] = 0u;
    __syncthreads();
    if (threadIdx.x == 0) floyd_subset(bits, nf, num_fg, rng, offset, 4u * b + 2u);
    __syncthreads();
    for (int i = threadIdx.x; i < nf; i += blockDim.x)
      if (!((bits[i >> 5] >> (i & 31)) & 1u)) lab[fg_list[(long)b * total + i]] = -1.f;
    __syncthreads();
  }
  const int fg_after = min(nf, num_fg);
  const int num_bg = batchsize - fg_after;
  if (nb > num_bg) {  // (:148-154)
    for (int i = threadIdx.x; i < words; i += blockDim.x) bits[i] = 0u;
    __syncthreads();
    if (threadIdx.x == 0) floyd_subset(bits, nb, num_bg, rng, offset, 4u * b + 3u);
    __syncthreads();
    for (int i = threadIdx.x; i < nb; i += blockDim.x)
      if (!((bits[i >> 5] >> (i & 31)) & 1u)) lab[bg_list[(long)b * total + i]] = -1.f;
  }
  // the LAST image's example count weights every image's box loss (:156, :176-177)
  if (b == B - 1 && threadIdx.x == 0) {
    const int ne = fg_after + min(nb, num_bg);
    inv_num_examples[0] = 1.0f / (float)max(ne, 1);
  }
}

}  // namespace

extern "C" {

static int proposal_target_sample_impl(const int* counts, int B, int n_candidates, int rois_per_image,
                                       int fg_rois_per_image, unsigned long long seed, unsigned long long offset,
                                       const unsigned long long* offset_dev, int* picks, int* fg_taken,
                                       dana_stream_t stream) {
  DANA_CHECK_ARG(B >= 0 && n_candidates > 0 && rois_per_image > 0 && fg_rois_per_image >= 0,
                 "dana_proposal_target_sample: bad shape");
  if (B == 0) return DANA_OK;
  DANA_CHECK_ARG(counts && picks && fg_taken, "dana_proposal_target_sample: null pointer");
  const size_t lds = (size_t)((n_candidates + 31) / 32) * 2 * sizeof(unsigned);
  DANA_CHECK_ARG(lds <= 60 * 1024, "dana_proposal_target_sample: too many candidates");
  proposal_target_sample_kernel<<<B, 256, lds, (hipStream_t)stream>>>(counts, n_candidates, rois_per_image,
                                                                      fg_rois_per_image, seed, offset, offset_dev,
                                                                      picks, fg_taken);
  DANA_CHECK_LAUNCH("dana_proposal_target_sample");
  return DANA_OK;
}

int dana_proposal_target_sample(const int* counts, int B, int n_candidates, int rois_per_image, int fg_rois_per_image,
                                unsigned long long seed, unsigned long long offset, int* picks, int* fg_taken,
                                dana_stream_t stream) {
  return proposal_target_sample_impl(counts, B, n_candidates, rois_per_image, fg_rois_per_image, seed, offset, nullptr,
                                     picks, fg_taken, stream);
}

int dana_proposal_target_sample_ctr(const int* counts, int B, int n_candidates, int rois_per_image,
                                    int fg_rois_per_image, unsigned long long seed, unsigned long long offset,
                                    const unsigned long long* counter_dev, int* picks, int* fg_taken,
                                    dana_stream_t stream) {
  DANA_CHECK_ARG(counter_dev, "dana_proposal_target_sample_ctr: null counter");
  return proposal_target_sample_impl(counts, B, n_candidates, rois_per_image, fg_rois_per_image, seed, offset,
                                     counter_dev, picks, fg_taken, stream);
}

int dana_counter_add(unsigned long long* counter_dev, unsigned long long inc, dana_stream_t stream) {
  DANA_CHECK_ARG(counter_dev, "dana_counter_add: null pointer");
  counter_add_kernel<<<1, 64, 0, (hipStream_t)stream>>>(counter_dev, inc);
  DANA_CHECK_LAUNCH("dana_counter_add");
  return DANA_OK;
}

static int anchor_target_subsample_impl(float* labels, const int* fg_list, const int* bg_list, const int* counts, int B,
                                        int anchors_per_image, int rpn_batchsize, int num_fg, unsigned long long seed,
                                        unsigned long long offset, const unsigned long long* offset_dev,
                                        float* inv_num_examples, dana_stream_t stream) {
  DANA_CHECK_ARG(B > 0 && anchors_per_image > 0 && rpn_batchsize > 0 && num_fg >= 0 && num_fg <= rpn_batchsize,
                 "dana_anchor_target_subsample: bad shape");
  DANA_CHECK_ARG(labels && fg_list && bg_list && counts && inv_num_examples, "dana_anchor_target_subsample: null pointer");
  const size_t lds = (size_t)((anchors_per_image + 31) / 32) * sizeof(unsigned);
  DANA_CHECK_ARG(lds <= 60 * 1024, "dana_anchor_target_subsample: too many anchors per image");
  anchor_target_subsample_kernel<<<B, 1024, lds, (hipStream_t)stream>>>(labels, fg_list, bg_list, counts, B,
                                                                        anchors_per_image, rpn_batchsize, num_fg, seed,
                                                                        offset, offset_dev, inv_num_examples);
  DANA_CHECK_LAUNCH("dana_anchor_target_subsample");
  return DANA_OK;
}

int dana_anchor_target_subsample(float* labels, const int* fg_list, const int* bg_list, const int* counts, int B,
                                 int anchors_per_image, int rpn_batchsize, int num_fg, unsigned long long seed,
                                 unsigned long long offset, float* inv_num_examples, dana_stream_t stream) {
  return anchor_target_subsample_impl(labels, fg_list, bg_list, counts, B, anchors_per_image, rpn_batchsize, num_fg, seed,
                                      offset, nullptr, inv_num_examples, stream);
}

int dana_anchor_target_subsample_ctr(float* labels, const int* fg_list, const int* bg_list, const int* counts, int B,
                                     int anchors_per_image, int rpn_batchsize, int num_fg, unsigned long long seed,
                                     unsigned long long offset, const unsigned long long* counter_dev,
                                     float* inv_num_examples, dana_stream_t stream) {
  DANA_CHECK_ARG(counter_dev, "dana_anchor_target_subsample_ctr: null counter");
  return anchor_target_subsample_impl(labels, fg_list, bg_list, counts, B, anchors_per_image, rpn_batchsize, num_fg, seed,
                                      offset, counter_dev, inv_num_examples, stream);
}

}  // extern "C"
