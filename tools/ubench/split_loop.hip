// ablation of the split-igemm K-step: which of {fragment reads, split VALU, staging writes, barrier} costs the time
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int V> struct IC { static constexpr int value = V; };
template <int I, int N, class F> __device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) { f(IC<I>{}); static_for<I + 1, N>(f); }
}
constexpr int SLD = 12, BM = 128, BN = 128, TM = 2, TN = 2, RA = 2, RB = 2;

template <int RD, int CV, int WR, int BAR, int ILV>
__global__ void __launch_bounds__(256, 1) k(float* out, const float* in, int iters) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  unsigned* As = (unsigned*)smem;
  unsigned* Bs = As + 2 * 3 * BM * SLD;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
  const int c4 = tid & 3, r0 = tid >> 2;
  for (int i = tid; i < 2 * 3 * (BM + BN) * SLD; i += 256) As[i] = 0;
  __syncthreads();
  f32x16 acc[TM][TN];
  for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  u32x4 fa0[3][TM], fb0[3][TN], fa1[3][TM], fb1[3][TN];
  for (int pc = 0; pc < 3; ++pc) for (int i = 0; i < 2; ++i) { fa0[pc][i] = fb0[pc][i] = fa1[pc][i] = fb1[pc][i] = *(const u32x4*)(in + tid * 4); }
  float4 cva[2], cvb[2];
  cva[0] = *(const float4*)(in + tid * 4); cva[1] = *(const float4*)(in + 1024 + tid * 4);
  cvb[0] = *(const float4*)(in + 2048 + tid * 4); cvb[1] = *(const float4*)(in + 3072 + tid * 4);
  constexpr int NM = 24, NF = 4, NR = 12, I_RD = 0, I_CV = NR, NI = I_CV + 7 * NF;
  auto k_step = [&](int t, const u32x4(&fa)[3][TM], const u32x4(&fb)[3][TN], u32x4(&na)[3][TM], u32x4(&nb)[3][TN]) {
    const int bw_ = t & 1, br_ = bw_ ^ 1;
    const unsigned* as = As + br_ * 3 * BM * SLD + (wm * (BM / 2) + li) * SLD + lh * 4;
    const unsigned* bs = Bs + br_ * 3 * BN * SLD + (wn * (BN / 2) + li) * SLD + lh * 4;
    unsigned* aw = As + bw_ * 3 * BM * SLD + r0 * SLD + c4 * 2;
    unsigned* bw = Bs + bw_ * 3 * BN * SLD + r0 * SLD + c4 * 2;
    constexpr int PA[6] = {0, 0, 1, 1, 0, 2};
    constexpr int PB[6] = {0, 1, 0, 1, 2, 0};
    unsigned hb[NF][4], mb[NF][4], lb[NF][4];
    uint2 hp[NF], mp[NF], lp[NF];
    __builtin_amdgcn_sched_barrier(0);
    static_for<0, NM>([&](auto qc) {
      constexpr int q = decltype(qc)::value;
      constexpr int pq = q / (TM * TN), ti = (q / TN) % TM, tj = q % TN;
      acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[PA[pq]][ti]),
                                                            __builtin_bit_cast(bf16x8, fb[PB[pq]][tj]), acc[ti][tj], 0, 0, 0);
      static_for<0, NI>([&](auto ic) {
        constexpr int it = decltype(ic)::value;
        if constexpr (it * NM / NI != q) {
        } else if constexpr (it < I_CV) {
          if constexpr (RD) {
            constexpr int r = it - I_RD;
            constexpr int pc = r / (TM + TN), x = r % (TM + TN);
            if constexpr (x < TM) na[pc][x < TM ? x : 0] = *(const u32x4*)(as + (pc * BM + x * 32) * SLD);
            else nb[pc][x < TM ? 0 : x - TM] = *(const u32x4*)(bs + (pc * BN + (x - TM) * 32) * SLD);
          }
        } else {
          constexpr int f = (it - I_CV) / 7, r = (it - I_CV) % 7;
          const float4 v = f < RA ? cva[f < RA ? f : 0] : cvb[f < RA ? 0 : f - RA];
          if constexpr (r == 0 || r == 3) {
            if constexpr (CV && ILV) {  // two elements' chains interleaved
              constexpr int e = r == 0 ? 0 : 2;
              float x0 = e == 0 ? v.x : v.z, x1 = e == 0 ? v.y : v.w;
              asm volatile("" : "+v"(x0), "+v"(x1));
              hb[f][e] = __float_as_uint(x0) & 0xffff0000u;
              hb[f][e + 1] = __float_as_uint(x1) & 0xffff0000u;
              const float r10 = x0 - __uint_as_float(hb[f][e]);
              const float r11 = x1 - __uint_as_float(hb[f][e + 1]);
              mb[f][e] = __float_as_uint(r10) & 0xffff0000u;
              mb[f][e + 1] = __float_as_uint(r11) & 0xffff0000u;
              lb[f][e] = __float_as_uint(r10 - __uint_as_float(mb[f][e]));
              lb[f][e + 1] = __float_as_uint(r11 - __uint_as_float(mb[f][e + 1]));
              asm volatile("" : "+v"(hb[f][e]), "+v"(mb[f][e]), "+v"(lb[f][e]), "+v"(hb[f][e + 1]), "+v"(mb[f][e + 1]), "+v"(lb[f][e + 1]));
            }
          }
          if constexpr (r == 0 || r == 1 || r == 3 || r == 4) {
            if constexpr (CV && !ILV) {
              constexpr int e = r < 2 ? r : r - 1;
              float x = e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w;
              asm volatile("" : "+v"(x));
              hb[f][e] = __float_as_uint(x) & 0xffff0000u;
              const float r1 = x - __uint_as_float(hb[f][e]);
              mb[f][e] = __float_as_uint(r1) & 0xffff0000u;
              lb[f][e] = __float_as_uint(r1 - __uint_as_float(mb[f][e]));
              asm volatile("" : "+v"(hb[f][e]), "+v"(mb[f][e]), "+v"(lb[f][e]));
            }
          } else if constexpr (r == 2 || r == 5) {
            if constexpr (CV) {
              constexpr int e = r == 2 ? 0 : 2;
              asm volatile("" : "+v"(hb[f][e]), "+v"(hb[f][e + 1]));
              unsigned a = __builtin_amdgcn_perm(hb[f][e + 1], hb[f][e], 0x07060302u);
              unsigned b = __builtin_amdgcn_perm(mb[f][e + 1], mb[f][e], 0x07060302u);
              unsigned c = __builtin_amdgcn_perm(lb[f][e + 1], lb[f][e], 0x07060302u);
              asm volatile("" : "+v"(a), "+v"(b), "+v"(c));
              if (e == 0) { hp[f].x = a; mp[f].x = b; lp[f].x = c; } else { hp[f].y = a; mp[f].y = b; lp[f].y = c; }
            }
          } else if constexpr (r == 6) {
            if constexpr (WR) {
              if constexpr (!CV) { hp[f] = make_uint2(__float_as_uint(v.x), __float_as_uint(v.y)); mp[f] = hp[f]; lp[f] = hp[f]; }
              unsigned* w = f < RA ? aw + 64 * f * SLD : bw + 64 * (f - RA) * SLD;
              constexpr int rows = f < RA ? BM : BN;
              *(uint2*)(w + 0 * rows * SLD) = hp[f];
              *(uint2*)(w + 1 * rows * SLD) = mp[f];
              *(uint2*)(w + 2 * rows * SLD) = lp[f];
            } else if constexpr (CV) {
              asm volatile("" :: "v"(hp[f].x), "v"(hp[f].y), "v"(mp[f].x), "v"(mp[f].y), "v"(lp[f].x), "v"(lp[f].y));
            }
          }
        }
      });
      __builtin_amdgcn_sched_barrier(0);
    });
    if (BAR) __syncthreads();
  };
  for (int t = 0; t < iters; t += 2) {
    k_step(t, fa0, fb0, fa1, fb1);
    k_step(t + 1, fa1, fb1, fa0, fb0);
  }
  float s = 0.f;
  for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[blockIdx.x * 256 + tid] = s;
}

template <int RD, int CV, int WR, int BAR, int ILV>
void run(int bpc, float* out, float* in) {
  const int iters = 2000;
  const size_t lds = 2 * 3 * (BM + BN) * SLD * 4;
  hipFuncSetAttribute((const void*)k<RD, CV, WR, BAR, ILV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<RD, CV, WR, BAR, ILV><<<256 * bpc, 256, lds>>>(out, in, 10);
  hipEventRecord(e0);
  k<RD, CV, WR, BAR, ILV><<<256 * bpc, 256, lds>>>(out, in, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("reads=%d split=%d writes=%d barrier=%d interleaved=%d blocks/CU=%d : %7.0f cycles@2.4GHz per K-step per wave (MFMA floor 768)\n",
         RD, CV, WR, BAR, ILV, bpc, ms * 1e-3 * 2.4e9 / iters);
}
int main() {
  float *out, *in; hipMalloc(&out, 512 * 256 * 4); hipMalloc(&in, 1 << 20); hipMemset(in, 0, 1 << 20);
  run<0, 0, 0, 0, 0>(1, out, in);
  run<1, 0, 0, 0, 0>(1, out, in);
  run<0, 1, 0, 0, 0>(1, out, in);
  run<0, 1, 0, 0, 1>(1, out, in);
  run<0, 0, 1, 0, 0>(1, out, in);
  run<0, 0, 0, 1, 0>(1, out, in);
  run<1, 1, 1, 1, 0>(1, out, in);
  run<1, 1, 1, 1, 1>(1, out, in);
  run<1, 1, 1, 1, 0>(2, out, in);
  run<1, 1, 1, 1, 1>(2, out, in);
  run<1, 0, 1, 1, 0>(1, out, in);
  run<1, 1, 0, 1, 1>(1, out, in);
  return 0;
}
