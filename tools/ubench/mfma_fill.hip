// micro-benchmark: v_mfma_f32_32x32x16_bf16 stream with V plain VALU fillers per MFMA, W waves per SIMD
// build: hipcc -O3 --offload-arch=gfx950 mfma_fill.hip -o mfma_fill ; run: ./mfma_fill
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int V, int DEP, int NACC>
__global__ void __launch_bounds__(256) k(float* out, const unsigned* in, int iters) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  u32x4 a = *(const u32x4*)(in + threadIdx.x * 4), b = *(const u32x4*)(in + 1024 + threadIdx.x * 4);
  unsigned x[8];
  for (int i = 0; i < 8; ++i) x[i] = in[threadIdx.x + i * 64];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < 24; ++q) {
      acc[q % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b),
                                                             acc[q % NACC], 0, 0, 0);
#pragma unroll
      for (int v = 0; v < V; ++v) {
        const int idx = DEP ? 0 : (v % 8);
        // alternate and / fsub like the split does
        if (v & 1) { float f = __uint_as_float(x[idx]) - 1.5f; x[idx] = __float_as_uint(f); }
        else x[idx] = x[idx] & 0xffff0fffu;
        asm volatile("" : "+v"(x[idx]));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int i = 0; i < 8; ++i) s += __uint_as_float(x[i]);
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int V, int DEP, int NACC>
void run(int blocks_per_cu, float* out, unsigned* in) {
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<V, DEP, NACC><<<256 * blocks_per_cu, 256>>>(out, in, 10);
  hipEventRecord(e0);
  k<V, DEP, NACC><<<256 * blocks_per_cu, 256>>>(out, in, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double cyc = ms * 1e-3 * 2.4e9 / iters / 24 / blocks_per_cu;
  printf("V=%d dep=%d nacc=%d waves/SIMD=%d : %.1f cycles per MFMA per wave (%.1f per SIMD-MFMA)\n", V, DEP, NACC, blocks_per_cu,
         cyc * blocks_per_cu, cyc);
}

int main() {
  float* out; unsigned* in;
  hipMalloc(&out, 256 * 4 * 256 * 4 * 4); hipMalloc(&in, 1 << 20); hipMemset(in, 0, 1 << 20);
  run<0, 0, 4>(1, out, in); run<1, 0, 4>(1, out, in); run<2, 0, 4>(1, out, in); run<3, 0, 4>(1, out, in);
  run<4, 0, 4>(1, out, in); run<5, 0, 4>(1, out, in); run<6, 0, 4>(1, out, in); run<8, 0, 4>(1, out, in);
  run<4, 1, 4>(1, out, in); run<6, 1, 4>(1, out, in);
  run<0, 0, 1>(1, out, in); run<0, 0, 2>(1, out, in);
  run<0, 0, 4>(2, out, in); run<4, 0, 4>(2, out, in); run<6, 0, 4>(2, out, in); run<8, 0, 4>(2, out, in);
  return 0;
}
