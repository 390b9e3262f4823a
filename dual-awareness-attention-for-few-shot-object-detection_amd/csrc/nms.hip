// Greedy NMS entirely on device (no D2H mask copy, no host scan) for gfx950.
//
// Replaces the reference's lib/model/csrc/cuda/nms.cu:
//   nms_kernel (:23-67)   -> nms_mask_kernel : 64x64 IoU tiles -> 64-bit suppression words, upper
//                            triangle only (the lower one is never read); one wave per tile, the
//                            grid enumerates (column group, row block) so every workgroup has the
//                            same amount of work (the reference's row-major sweep is triangular).
//   host scan (:100-123)  -> nms_scan_flow_kernel / nms_scan_kernel : one 1024-thread workgroup per problem walks the
//                            64-box blocks in order. Wave 0 (the resolver) settles block b as a FIXED POINT
//                            instead of the reference's box-by-box chain: lane j holds the bits of the earlier
//                            boxes of its block that suppress box j (the transposed diagonal tile, written by
//                            the mask kernel), so "kept = alive and no kept earlier box suppresses me" is one
//                            ballot per round, K <- ballot(alive_j && (col_j & K) == 0). Suppression only points
//                            from earlier to later boxes, so the rounds settle the boxes in order of their
//                            dependency depth and the fixed point IS the greedy result (1.4 rounds per block on the
//                            proposal layer's boxes, where the scalar chain cost ~165 cycles per kept box). The kept
//                            boxes of the last block(s) enter the same way through the transposed tiles of the first
//                            super-diagonal(s); the other waves OR the kept rows of earlier blocks into the LDS-resident
//                            `remv`. Two forms: the barrier-free dataflow kernel (bands of <= 128 column blocks: every
//                            band of the proposal layer; described at the kernel) and the round 2-4 kernel with one
//                            workgroup barrier per block (wider bands).
// Column bands when the caller keeps at most `max_keep` boxes (the RPN's post_nms_topN): greedy NMS stops after max_keep
// keeps, which on score-sorted proposals happens long before the last box, and the scan only ever reads mask words
// of rows it has visited. Band 0 fills and scans the triangle of the first ~2.5 * max_keep boxes; every later band
// (its column band of the mask, then its stretch of the scan) leaves at once if an earlier one reached max_keep. A
// band's tiles fold the rows kept so far into a per-column "already suppressed" word on the way (one wave OR + one
// atomic per tile), so its scan starts from the exact state a single full scan would have at that block.
// IoU uses the legacy "+1" widths (nms.cu:13-21). `inclusive`==0 suppresses on IoU > thr
// (reference CUDA, nms.cu:60); ==1 on IoU >= thr (reference CPU, cpu/nms_cpu.cpp:60).
// Compiled with -ffp-contract=off. The division-free fast path below only decides cases that are
// at least 2^-21 (relative) away from the threshold; anything closer takes the exact IEEE division,
// so every decision equals the reference's `inter / (Sa + Sb - inter) > thr` bit for bit.
#include "common.h"
#include "../../include/dana_hip.h"
#include <stdlib.h>

namespace {

__device__ __forceinline__ bool iou_suppress(float4 a, float sa, float4 b, float sb, float thr, int inclusive) {
  const float left = fmaxf(a.x, b.x), right = fminf(a.z, b.z);
  const float top = fmaxf(a.y, b.y), bottom = fminf(a.w, b.w);
  const float w = fmaxf(right - left + 1.f, 0.f), h = fmaxf(bottom - top + 1.f, 0.f);
  const float inter = w * h;
  const float uni = sa + sb - inter;
  const float t = thr * uni;
  if (uni > 0.f && thr > 0.f) {
    if (inter > t * (1.f + 4.8e-7f)) return true;   // certainly above the threshold
    if (inter < t * (1.f - 4.8e-7f)) return false;  // certainly below
  }
  const float ovr = inter / uni;
  return inclusive ? (ovr >= thr) : (ovr > thr);
}

__device__ __forceinline__ float box_area(float4 a) { return (a.z - a.x + 1.f) * (a.w - a.y + 1.f); }

__device__ __forceinline__ unsigned long long wave_or64(unsigned long long v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v |= __shfl_xor(v, o);
  return v;
}

// per-problem hand-off between the passes (workspace): state[0] = keeps so far, state[1] = finished
struct NmsPass {
  int cb_lo, cb_hi;                      // column blocks of this launch
  const int* state;                      // pass 2: leave if state[2 * problem + 1]
  const unsigned long long* keptmask;    // pass 2: bit r of keptmask[problem][rb] = row rb*64+r was kept in pass 1
  unsigned long long* remv_g;            // pass 2: OR of the kept rows' words, per column block
  int kept_rows_hi;                      // row blocks < this were scanned in pass 1
  unsigned long long* diag_t;            // [problem][n]: bits of the EARLIER boxes of box j's block that suppress j
  unsigned long long* prev_t;            // [problem][n]: bits of the boxes of the PREVIOUS block that suppress j
  unsigned long long* prev2_t;           // [problem][n]: ... of the block before that
  unsigned long long* prev3_t;           // [problem][n]: ... and of the one before that
};

// grid = (ceil((cb_hi - cb_lo)/4), cb_hi, problems); wave w of the workgroup owns tile (rb, cb_lo + cg*4 + w)
__global__ void __launch_bounds__(256)
nms_mask_kernel(const float4* __restrict__ boxes, unsigned long long* __restrict__ mask, int n, int col_blocks,
                float thr, int inclusive, NmsPass ps) {
  __shared__ float4 colbox[4][64];
  __shared__ float colarea[4][64];
  if (ps.state && ps.state[2 * blockIdx.z + 1]) return;  // pass 1 already kept max_keep boxes
  const int rb = blockIdx.y;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int cb = ps.cb_lo + blockIdx.x * 4 + wave;
  if (cb + 3 < rb || cb >= ps.cb_hi) return;  // wave-uniform: whole waves leave, no barrier below
  const float4* pb = boxes + (long)blockIdx.z * n;
  unsigned long long* pm = mask + (long)blockIdx.z * n * col_blocks;
  const int row = rb * 64 + lane;
  const float4 me = row < n ? pb[row] : make_float4(0, 0, 0, 0);
  const float sme = box_area(me);
  const int col = cb * 64 + lane;
  const float4 cbx = col < n ? pb[col] : make_float4(0, 0, 0, 0);
  colbox[wave][lane] = cbx;
  colarea[wave][lane] = box_area(cbx);
  __builtin_amdgcn_wave_barrier();
  if (cb < rb) {
    // One of the three tiles left of the diagonal: not a mask tile (the lower triangle is never read) but the TRANSPOSED
    // tile of super-diagonal rb - cb -- bit i of box j's word = box i of the earlier block cb suppresses box j (the test is
    // symmetric bit for bit: max / min / the commutative sa + sb). These waves would leave at once otherwise.
    unsigned long long w = 0;
    for (int i = 0; i < 64; ++i)  // (an earlier block is always full)
      if (iou_suppress(me, sme, colbox[wave][i], colarea[wave][i], thr, inclusive)) w |= 1ULL << i;
    unsigned long long* dst = rb - cb == 1 ? ps.prev_t : (rb - cb == 2 ? ps.prev2_t : ps.prev3_t);
    if (row < n) dst[(long)blockIdx.z * n + row] = w;
    return;
  }
  const int csize = min(64, n - cb * 64);
  unsigned long long t = 0;
  if (cb != rb) {
    for (int i = 0; i < csize; ++i)
      if (iou_suppress(me, sme, colbox[wave][i], colarea[wave][i], thr, inclusive)) t |= 1ULL << i;
  } else {
    // Diagonal tile: the test is symmetric bit for bit (max / min / the commutative sa + sb), so the same sweep over
    // ALL boxes of the block gives the row word (later boxes this one suppresses) and its transpose (earlier boxes
    // that suppress this one); three more sweeps, over the three blocks before this one, give the transposed tiles of
    // the first three super-diagonals.
    for (int i = 0; i < csize; ++i)
      if (i != lane && iou_suppress(me, sme, colbox[wave][i], colarea[wave][i], thr, inclusive)) t |= 1ULL << i;
    const unsigned long long below = (1ULL << lane) - 1;
    if (row < n) ps.diag_t[(long)blockIdx.z * n + row] = t & below;
    // the transposed super-diagonal tiles whose column block lies left of this band (no wave of this grid has them; a
    // band's first three diagonal blocks) -- and the zero words of the blocks that have no such earlier block
    for (int back = 1; back <= 3; ++back) {
      if (rb - back >= ps.cb_lo) continue;
      unsigned long long w = 0;
      if (rb - back >= 0) {
        __builtin_amdgcn_wave_barrier();
        const float4 pbx = pb[(rb - back) * 64 + lane];
        colbox[wave][lane] = pbx;
        colarea[wave][lane] = box_area(pbx);
        __builtin_amdgcn_wave_barrier();
        for (int i = 0; i < 64; ++i)
          if (iou_suppress(me, sme, colbox[wave][i], colarea[wave][i], thr, inclusive)) w |= 1ULL << i;
      }
      unsigned long long* dst = back == 1 ? ps.prev_t : (back == 2 ? ps.prev2_t : ps.prev3_t);
      if (row < n) dst[(long)blockIdx.z * n + row] = w;
    }
    t &= ~below;
  }
  if (row < n) pm[(long)row * col_blocks + cb] = t;
  if (ps.keptmask && rb < ps.kept_rows_hi) {
    const unsigned long long kept = ps.keptmask[(long)blockIdx.z * col_blocks + rb];
    const unsigned long long v = wave_or64(((kept >> lane) & 1ULL) && row < n ? t : 0ULL);
    if (lane == 0 && v) atomicOr(&ps.remv_g[(long)blockIdx.z * col_blocks + cb], v);
  }
}

__device__ __forceinline__ unsigned long long uniform64(unsigned long long v) {
  return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32)) << 32) |
         (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
}

// LDS: remv[col_blocks] | list[2][64] ints | cnt[2] | count | done
// Scans the blocks [b_lo, b_hi) of every problem. b_lo == 0: fresh start (and the hand-off words are reset);
// b_lo > 0: continues from the state pass 1 left (count, per-column words folded by the band's mask tiles).
__global__ void __launch_bounds__(1024)
nms_scan_kernel(const unsigned long long* __restrict__ mask, int n, int col_blocks, int max_keep,
                int* __restrict__ keep, int* __restrict__ num_keep, int keep_stride, int b_lo, int b_hi,
                int* __restrict__ state, unsigned long long* __restrict__ keptmask,
                unsigned long long* __restrict__ remv_g, const unsigned long long* __restrict__ diag_t,
                const unsigned long long* __restrict__ prev_t) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long remv[];
  int* list = (int*)(remv + col_blocks);  // [2][64] kept rows (local index) of the last two blocks
  int* cnt = list + 128;                  // [2]
  int* s_count = cnt + 2;
  int* s_done = cnt + 3;
  const unsigned long long* pm = mask + (long)blockIdx.x * n * col_blocks;
  int* pk = keep + (long)blockIdx.x * keep_stride;
  int* st = state ? state + 2 * blockIdx.x : nullptr;
  unsigned long long* km = keptmask ? keptmask + (long)blockIdx.x * col_blocks : nullptr;
  unsigned long long* rg = remv_g ? remv_g + (long)blockIdx.x * col_blocks : nullptr;
  if (b_lo > 0 && st[1]) return;  // pass 1 finished the problem (num_keep is already written)
  for (int j = threadIdx.x; j < col_blocks; j += blockDim.x) {
    remv[j] = b_lo > 0 ? rg[j] : 0ULL;
    if (b_lo == 0 && rg) rg[j] = 0ULL;  // the band's mask tiles OR into it before pass 2 reads it
  }
  if (threadIdx.x < 4) cnt[threadIdx.x] = 0;  // cnt[0..1], count, done
  __syncthreads();
  if (threadIdx.x == 0 && b_lo > 0) *s_count = st[0];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // resolver prefetch registers: the transposed diagonal / super-diagonal words of the boxes of block b
  const unsigned long long* dt = diag_t + (long)blockIdx.x * n;
  const unsigned long long* pt = prev_t + (long)blockIdx.x * n;
  unsigned long long colw = 0, prvw = 0;
  unsigned long long kprev = 0;  // kept boxes of block b-1 (uniform); at the start of a band they are already in remv
  if (wave == 0) {
    const int row0 = b_lo * 64 + lane;
    if (row0 < n) {
      colw = dt[row0];
      prvw = pt[row0];
    }
  }
  for (int b = b_lo; b < b_hi; ++b) {
    if (wave == 0) {
      // ---- resolver: prefetch block b+1's words first so their latency hides under the resolve ----
      unsigned long long colw_n = 0, prvw_n = 0;
      const int rown = (b + 1) * 64 + lane;
      if (b + 1 < b_hi && rown < n) {
        colw_n = dt[rown];
        prvw_n = pt[rown];
      }
      const int bsize = min(64, n - b * 64);
      const unsigned long long rm = uniform64(remv[b]);
      const bool al = lane < bsize && !((rm >> lane) & 1ULL) && (prvw & kprev) == 0ULL;
      unsigned long long K = __ballot(al);
      for (;;) {  // fixed point of K = alive & ~suppressed_by(K): the greedy result (see the header)
        const unsigned long long Kn = __ballot(al && (colw & K) == 0ULL);
        if (Kn == K) break;
        K = Kn;
      }
      const int base = __builtin_amdgcn_readfirstlane(*s_count);
      const unsigned long long below = (1ULL << lane) - 1;
      int c = __builtin_popcountll(K);
      if (base + c > max_keep) {  // the block that reaches max_keep: its first max_keep - base kept boxes
        c = max_keep - base;
        K = __ballot(((K >> lane) & 1ULL) && __builtin_popcountll(K & below) < c);
      }
      const bool mine = (K >> lane) & 1ULL;
      const int pos = __builtin_popcountll(K & below);
      if (mine) {
        pk[base + pos] = b * 64 + lane;
        list[(b & 1) * 64 + pos] = lane;
      }
      if (lane == 0) {
        cnt[b & 1] = c;
        *s_count = base + c;
        if (base + c == max_keep) *s_done = 1;
        if (km) km[b] = K;
      }
      kprev = K;
      colw = colw_n;
      prvw = prvw_n;
    } else if (b > b_lo) {
      // ---- workers: rows kept in block b-1 -> remv[j], b+1 <= j < b_hi (one iteration behind the resolver) ----
      // The (kept row, group of 64 column words) items are dealt round-robin to the 15 worker waves and each wave issues
      // ALL its loads before it uses any (up to 6 in flight): one L2 round trip per block instead of one per column
      // group. With the resolver a fixed point (~600 cycles) this round trip is what an iteration costs: ~1.1 us per block
      // (round 1: 3.2 us, the scalar resolver chain: 2.5).
      const int c = cnt[(b - 1) & 1];
      const int* rows = list + ((b - 1) & 1) * 64;
      const unsigned long long* base = pm + (long)(b - 1) * 64 * col_blocks;
      const int first = b + 1;
      const int ngroups = (b_hi - first + 63) / 64;
      const int items = c * ngroups;
      constexpr int U = 6;
      for (int it0 = wave - 1; it0 < items; it0 += 15 * U) {
        unsigned long long v[U];
        int jj[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int it = it0 + u * 15;
          v[u] = 0;
          jj[u] = -1;
          if (it < items) {
            const int i = it / ngroups, g = it - i * ngroups;
            const int j = first + g * 64 + lane;
            if (j < b_hi) {
              jj[u] = j;
              v[u] = base[(long)rows[i] * col_blocks + j];
            }
          }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (jj[u] >= 0 && v[u]) atomicOr(&remv[jj[u]], v[u]);
      }
    }
    __syncthreads();
    if (*s_done) break;
  }
  if (threadIdx.x == 0) {
    num_keep[blockIdx.x] = *s_count;
    if (st) {
      st[0] = *s_count;
      st[1] = (*s_done || b_hi >= col_blocks) ? 1 : 0;
    }
  }
}

// ---- the scan for bands of <= 128 column blocks (every band of the proposal layer's problems): a dataflow of waves with no
// workgroup barrier and no memory round trip on the resolver's path. Measured on the kernel above: 1.27 us per block = one
// round trip to the words the mask kernel just wrote (another XCD's L2 / the memory-side cache), paid by the resolver (its
// prefetch of block b+1 is waited for at the end of the iteration: a register copy across the loop edge needs the data) and
// by the workers (kept rows of block b-1, loaded and OR-ed inside one iteration). Holding loads in registers ACROSS
// iterations does not help: the compiler's wait-count pass is conservative for loads that are live across a loop edge
// (vmcnt(0..3) where 12-24 younger loads are in flight) -- so here no load is live across an iteration of its wave:
//   wave 0        resolver: block after block, words from an LDS ring, K published to LDS, keep list stored directly;
//                 waits (LDS flags) only for its words and for "rows of block b-4 folded"
//   waves 1-2     word loaders: the transposed diagonal / first three super-diagonal words of 4 blocks per turn -> LDS ring
//                 of 16 blocks, up to 12 blocks ahead of the resolver
//   waves 3-14    row workers, 4 groups x 3 waves, block b -> group b % 4: load ALL 64 rows' words for the column blocks
//                 >= b+4 (before the resolver gets there: nothing to wait for), then wait for K_b, OR the kept rows into
//                 remv, count the block as folded. The resolver reaches the kept boxes of the last three blocks through the
//                 transposed super-diagonal tiles; a group has four resolver steps per block (one memory round trip + the fold).
// LDS: remv[col_blocks] | ring[16][4][64] | kw[128] | folded[128] | wflag[4] | count, done, resolved, stop
constexpr int NMS_FLOW_MAX_BLOCKS = 128, NMS_FLOW_RING = 16;
constexpr int NMS_FLOW_GROUPS = 4, NMS_FLOW_GW = 3, NMS_FLOW_U = 22;  // 4 worker groups x 3 waves x 22 rows each (>= 64;
                                                                       // 6 x 2 x 32 spills: 128 registers per lane)
__device__ __forceinline__ void lds_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// LDS flags between the waves of one workgroup: relaxed workgroup-scope atomics (re-read on every poll, no fence) -- a
// `volatile` access makes the compiler wait for ALL outstanding memory operations (vmcnt(0): the resolver's keep-list stores,
// ~1 500 cycles per block)
template <typename T> __device__ __forceinline__ T flag_ld(const T* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
template <typename T> __device__ __forceinline__ void flag_st(T* p, T v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

__global__ void __launch_bounds__(1024)
nms_scan_flow_kernel(const unsigned long long* __restrict__ mask, int n, int col_blocks, int max_keep,
                     int* __restrict__ keep, int* __restrict__ num_keep, int keep_stride, int b_lo, int b_hi,
                     int* __restrict__ state, unsigned long long* __restrict__ keptmask,
                     unsigned long long* __restrict__ remv_g, const unsigned long long* __restrict__ diag_t,
                     const unsigned long long* __restrict__ prev_t, const unsigned long long* __restrict__ prev2_t,
                     const unsigned long long* __restrict__ prev3_t) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long remv[];
  unsigned long long* ring = remv + col_blocks;                          // [16 blocks][4 tables][64 lanes]
  unsigned long long* kw = ring + NMS_FLOW_RING * 4 * 64;       // [128] kept boxes of block b_lo + i
  int* folded = (int*)(kw + NMS_FLOW_MAX_BLOCKS);      // [128] worker waves done with block b_lo + i
  int* wflag = folded + NMS_FLOW_MAX_BLOCKS;                    // [4] chunk index + 1 held by ring quarter i
  int* s_count = wflag + 4;
  int* s_done = wflag + 5;
  int* s_res = wflag + 6;   // blocks resolved so far (relative to b_lo)
  int* s_stop = wflag + 7;
  const unsigned long long* pm = mask + (long)blockIdx.x * n * col_blocks;
  int* pk = keep + (long)blockIdx.x * keep_stride;
  int* st = state ? state + 2 * blockIdx.x : nullptr;
  unsigned long long* km = keptmask ? keptmask + (long)blockIdx.x * col_blocks : nullptr;
  unsigned long long* rg = remv_g ? remv_g + (long)blockIdx.x * col_blocks : nullptr;
  if (b_lo > 0 && st[1]) return;  // pass 1 finished the problem (num_keep is already written)
  for (int j = threadIdx.x; j < col_blocks; j += blockDim.x) {
    remv[j] = b_lo > 0 ? rg[j] : 0ULL;
    if (b_lo == 0 && rg) rg[j] = 0ULL;  // the band's mask tiles OR into it before pass 2 reads it
  }
  if (threadIdx.x < NMS_FLOW_MAX_BLOCKS) folded[threadIdx.x] = 0;
  if (threadIdx.x < 8) wflag[threadIdx.x] = 0;
  __syncthreads();
  if (threadIdx.x == 0 && b_lo > 0) *s_count = st[0];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nblk = b_hi - b_lo;
  if (wave == 0) {
    // ---- resolver ----
    unsigned long long k1 = 0, k2 = 0, k3 = 0;  // kept boxes of blocks b-1 .. b-3 (at the start of a band: already in remv)
    int base = flag_ld(s_count);                // keeps so far (this wave is the only writer)
    for (int i = 0; i < nblk; ++i) {
      const int b = b_lo + i;
      if ((i & 3) == 0) {  // a new chunk of the ring
        while (flag_ld(&wflag[(i >> 2) & 3]) != (i >> 2) + 1) __builtin_amdgcn_s_sleep(1);
        lds_wait();
      }
      // one LDS round trip for everything the block needs: its words, the fold counter of block b-4, the folded rows
      const unsigned long long* w = ring + (i & (NMS_FLOW_RING - 1)) * 256 + lane;
      const unsigned long long colw = w[0], prvw = w[64], prv2w = w[128], prv3w = w[192];
      // (the counter is read BEFORE the word -- one wave's LDS reads execute in order -- so a complete count means a
      //  complete word; late workers: poll the counter, then take the word again)
      int fold = i >= 4 ? flag_ld(&folded[i - 4]) : NMS_FLOW_GW;
      asm volatile("" ::: "memory");  // (the compiler keeps the order of the two reads; the LDS executes them in order)
      unsigned long long rm = flag_ld(&remv[b]);  // rows kept in blocks <= b-4 (and in earlier bands)
      lds_wait();
      while (fold < NMS_FLOW_GW) {
        __builtin_amdgcn_s_sleep(1);
        fold = flag_ld(&folded[i - 4]);
        asm volatile("" ::: "memory");
        rm = flag_ld(&remv[b]);
        lds_wait();
      }
      rm = uniform64(rm);
      const int bsize = min(64, n - b * 64);
      const bool al = lane < bsize && !((rm >> lane) & 1ULL) && (prvw & k1) == 0ULL && (prv2w & k2) == 0ULL &&
                      (prv3w & k3) == 0ULL;
      unsigned long long K = __ballot(al);
      for (;;) {  // fixed point of K = alive & ~suppressed_by(K): the greedy result (see the header)
        const unsigned long long Kn = __ballot(al && (colw & K) == 0ULL);
        if (Kn == K) break;
        K = Kn;
      }
      const unsigned long long below = (1ULL << lane) - 1;
      int c = __builtin_popcountll(K);
      if (base + c > max_keep) {  // the block that reaches max_keep: its first max_keep - base kept boxes
        c = max_keep - base;
        K = __ballot(((K >> lane) & 1ULL) && __builtin_popcountll(K & below) < c);
      }
      if ((K >> lane) & 1ULL) pk[base + __builtin_popcountll(K & below)] = b * 64 + lane;
      base += c;
      if (lane == 0) {
        flag_st(&kw[i], K);
        asm volatile("" ::: "memory");
        flag_st(s_res, i + 1);  // (behind kw[i]: one wave's LDS operations execute in order)
        if (km) km[b] = K;
      }
      k3 = k2;
      k2 = k1;
      k1 = K;
      if (base == max_keep) break;
    }
    if (lane == 0) {
      flag_st(s_count, base);
      if (base == max_keep) flag_st(s_done, 1);
      flag_st(s_stop, 1);
    }
  } else if (wave <= 2) {
    // ---- word loaders: chunk ci = blocks [4 ci, 4 ci + 4) of the band, chunks alternate between the two waves ----
    const unsigned long long* tab[4] = {diag_t + (long)blockIdx.x * n, prev_t + (long)blockIdx.x * n,
                                        prev2_t + (long)blockIdx.x * n, prev3_t + (long)blockIdx.x * n};
    for (int ci = wave - 1; ci * 4 < nblk; ci += 2) {
      // the ring quarter is free once the blocks that used it (chunk ci - 4) are resolved
      while (flag_ld(s_res) < ci * 4 - 12 && !flag_ld(s_stop)) __builtin_amdgcn_s_sleep(1);
      if (flag_ld(s_stop)) break;
      unsigned long long v[4][4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r0 = (b_lo + ci * 4 + q) * 64 + lane;
        const int r = (ci * 4 + q < nblk && r0 < n) ? r0 : 0;  // (row 0's words are zero: no earlier box, no earlier block)
#pragma unroll
        for (int t = 0; t < 4; ++t) v[q][t] = tab[t][r];
      }
      unsigned long long* dst = ring + ((ci * 4) & (NMS_FLOW_RING - 1)) * 256 + lane;
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int t = 0; t < 4; ++t) dst[q * 256 + t * 64] = v[q][t];
      lds_wait();
      if (lane == 0) flag_st(&wflag[ci & 3], ci + 1);
    }
  } else if (wave <= 14) {
    // ---- row workers ----
    const int grp = (wave - 3) / NMS_FLOW_GW, sub = (wave - 3) % NMS_FLOW_GW;
    constexpr int U = NMS_FLOW_U;
    for (int i = grp; i < nblk; i += NMS_FLOW_GROUPS) {
      const int b = b_lo + i;
      const int first = b + 4;
      bool stop = false;
      for (int g0 = first; g0 < b_hi && !stop; g0 += 64) {  // column groups of 64 blocks (two only at the head of a wide band)
        const int j = g0 + lane;
        unsigned long long hv[U];
        // (32-bit offsets from the uniform base: 64-bit lane addresses for 22 loads do not fit the 128 registers of a
        //  1024-lane workgroup; the mask of one problem is < 4 GB for every n the LDS-resident scan accepts)
        // every load is issued whatever the item (a row or column behind the end reads word 0 of the mask and is dropped
        // at the fold): a load under a condition is waited for at the join behind it -- 22 round trips in a row
        const unsigned off0 = (unsigned)(b * 64 + sub) * (unsigned)col_blocks + (unsigned)j;
        const bool colok = j < b_hi;
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int r = sub + NMS_FLOW_GW * u;
          const bool ok = r < 64 && b * 64 + r < n && colok;
          hv[u] = pm[ok ? off0 + (unsigned)(NMS_FLOW_GW * u) * (unsigned)col_blocks : 0u];
        }
        while (flag_ld(s_res) <= i && !flag_ld(s_stop)) __builtin_amdgcn_s_sleep(1);
        if (flag_ld(s_res) <= i) {  // the resolver stopped before this block
          stop = true;
          break;
        }
        lds_wait();
        const unsigned long long K = flag_ld(&kw[i]);
        unsigned long long acc = 0;
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (sub + NMS_FLOW_GW * u < 64 && ((K >> ((sub + NMS_FLOW_GW * u) & 63)) & 1ULL)) acc |= hv[u];  // (K has no bit for a row >= n)
        if (acc && colok) atomicOr(&remv[j], acc);
      }
      if (stop) break;
      lds_wait();
      if (lane == 0) atomicAdd(&folded[i], 1);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    num_keep[blockIdx.x] = *s_count;
    if (st) {
      st[0] = *s_count;
      st[1] = (*s_done || b_hi >= col_blocks) ? 1 : 0;
    }
  }
}

}  // namespace

extern "C" {

size_t dana_nms_workspace_bytes(int n, int problems) {
  if (n <= 0 || problems <= 0) return 0;
  size_t cb = (size_t)(n + 63) / 64;
  // mask words | kept-row words [problems][cb] | folded column words [problems][cb] | state [problems][2]
  // ... | transposed diagonal / first three super-diagonal words [4][problems][n]
  return (size_t)problems * n * cb * sizeof(unsigned long long) + (size_t)problems * (2 * cb * 8 + 16) +
         (size_t)4 * problems * n * sizeof(unsigned long long);
}

int dana_nms(const float* boxes, int n, int problems, float thr, int inclusive, int max_keep, int* keep,
             int keep_stride, int* num_keep, void* workspace, size_t workspace_bytes, dana_stream_t stream) {
  DANA_CHECK_ARG(n >= 0 && problems >= 0, "dana_nms: bad n=%d problems=%d", n, problems);
  hipStream_t s = (hipStream_t)stream;
  if (problems == 0) return DANA_OK;
  DANA_CHECK_ARG(num_keep, "dana_nms: null num_keep");
  if (n == 0) {  // nms.h:17-18: empty input -> empty result
    if (hipMemsetAsync(num_keep, 0, sizeof(int) * problems, s) != hipSuccess) {
      dana_set_error("dana_nms: memset failed");
      return DANA_ERR_HIP;
    }
    return DANA_OK;
  }
  DANA_CHECK_ARG(boxes && keep, "dana_nms: null pointer");
  if (max_keep <= 0 || max_keep > n) max_keep = n;
  DANA_CHECK_ARG(keep_stride >= max_keep, "dana_nms: keep_stride %d < max_keep %d", keep_stride, max_keep);
  DANA_CHECK_ARG(((uintptr_t)boxes & 15) == 0, "dana_nms: boxes must be 16-byte aligned");
  size_t need = dana_nms_workspace_bytes(n, problems);
  if (workspace_bytes < need || !workspace) {
    dana_set_error("dana_nms: workspace %zu < %zu", workspace_bytes, need);
    return DANA_ERR_WORKSPACE;
  }
  const int cb = (n + 63) / 64;
  const size_t lds = (size_t)cb * 8 + 128 * 4 + 16;
  const size_t flow_lds = (size_t)cb * 8 + NMS_FLOW_RING * 4 * 64 * 8 + NMS_FLOW_MAX_BLOCKS * 12 + 64;
  DANA_CHECK_ARG(lds <= 64 * 1024 && cb <= 65535, "dana_nms: n=%d too large for the LDS-resident scan", n);
  unsigned long long* maskw = (unsigned long long*)workspace;
  unsigned long long* keptmask = maskw + (size_t)problems * n * cb;
  unsigned long long* remv_g = keptmask + (size_t)problems * cb;
  int* state = (int*)(remv_g + (size_t)problems * cb);
  unsigned long long* diag_t = (unsigned long long*)(state + 2 * (size_t)problems + 2 * ((size_t)problems & 1));
  unsigned long long* prev_t = diag_t + (size_t)problems * n;
  unsigned long long* prev2_t = prev_t + (size_t)problems * n;
  unsigned long long* prev3_t = prev2_t + (size_t)problems * n;
  // Band edges in units of max_keep boxes x 100: the first pass covers 2.5 x max_keep boxes, the second up to 5 x, the last
  // the rest (measured best on the proposal layer's 12 000 -> 2 000 problems; a single pass is what "0" edges give). An
  // edge that would leave less than a quarter of the problem for the later bands is dropped.
  static const int edges_pct[2] = {250, 500};
  const int n_edges = 2;
  int edge[10], nb = 0;  // band i = column blocks [edge[i], edge[i+1])
  edge[nb++] = 0;
  // (worth it only when the full triangle is a real cost: a band that turns out to be needed adds ~25 us of restart)
  if (max_keep < n && (long)problems * cb * cb / 2 >= 20000)
    for (int i = 0; i < n_edges; ++i) {
      long blocks = ((long)max_keep * edges_pct[i] / 100 + 63) / 64;
      if (blocks < 32 * (i + 1)) blocks = 32 * (i + 1);  // a band's mask costs little below ~2000 boxes, a launch does
      if (blocks > edge[nb - 1] && blocks * 4 <= (long)cb * 3 && nb <= cb / 64) edge[nb++] = (int)blocks;  // <= 1 + cb/64 bands
    }
  edge[nb] = cb;
  for (int i = 0; i < nb; ++i) {
    const int lo = edge[i], hi = edge[i + 1];
    const NmsPass ps = {lo, hi, i ? state : nullptr, i ? keptmask : nullptr, i ? remv_g : nullptr, lo, diag_t, prev_t, prev2_t,
                       prev3_t};
    nms_mask_kernel<<<dim3((hi - lo + 3) / 4, hi, problems), 256, 0, s>>>((const float4*)boxes, maskw, n, cb, thr, inclusive,
                                                                          ps);
    DANA_CHECK_LAUNCH("dana_nms(mask)");
    if (hi - lo <= NMS_FLOW_MAX_BLOCKS && flow_lds <= 64 * 1024)
      nms_scan_flow_kernel<<<problems, 1024, flow_lds, s>>>(maskw, n, cb, max_keep, keep, num_keep, keep_stride, lo, hi, state,
                                                            keptmask, remv_g, diag_t, prev_t, prev2_t, prev3_t);
    else  // (a band wider than 128 column blocks: a single-band problem of more than 8 192 boxes)
      nms_scan_kernel<<<problems, 1024, lds, s>>>(maskw, n, cb, max_keep, keep, num_keep, keep_stride, lo, hi, state, keptmask,
                                                  remv_g, diag_t, prev_t);
    DANA_CHECK_LAUNCH("dana_nms(scan)");
  }
  return DANA_OK;
}

}  // extern "C"
