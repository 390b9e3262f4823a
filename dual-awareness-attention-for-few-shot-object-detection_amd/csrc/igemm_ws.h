// Warp-specialised persistent form of the split contraction (round 4) -- included by igemm.hip inside its namespace.
// STATUS: experimental, OFF by default (dana_set_ws_mode / DANA_WS). Bit-identical to igemm_split_kernel on every shape
// (tests/test_gpu_contractions.py::test_warp_specialised_kernel_gives_the_split_kernels_bits), and 1.03-1.9x its duration
// (profiles/r4_ws_sweep.md): parity on the long-K launches, slower on the short-K ones it was built for. What the role
// ablation showed: consumers alone hold 0.52 us per K-step (768 MFMA cycles = 0.41); the stagers' VALU / LDS
// work on out-of-range loads costs +0.05; REAL loads cost +0.35 -- four K-steps of loads in flight per stager thread
// (80 KB per CU, the split kernel's amount) do not cover the loaded latency when ONE tile per CU has to be fed at the
// matrix pipe's full rate; a ring of six K-steps (it fits the stagers' registers) measured the same.
//
// Why: a 128 x 128 tile of igemm_split_kernel spends 7 800 cycles before its first MFMA and 1 700 .. 14 500 behind its
// last one, against a K loop of 740 cycles per step and tile: at K = 256 the fixed phases are as long as the loop, and two
// workgroups per CU (224 registers, 67.6 KB LDS) run them in lock step, so the matrix pipe idles through both. Chaining
// launches cannot recover that; pipelining ACROSS the tiles of one launch can, if the roles are separated:
//
//   waves 0-3  consumers : fragment reads + the 24 MFMAs of a K-step, nothing else; at a tile's last step the accumulators
//                          go to an LDS C tile and the next tile's first MFMA follows at once
//   waves 4-7  stagers   : global loads (two K-steps ahead), the three-way bf16 split, LDS staging writes -- they run on
//                          into the NEXT tile's K-steps while the consumers finish the current one (no prologue)
//   waves 8-11 finishers : the previous tile's epilogue (C tile from LDS, scale / shift, residual, ReLU, float4 stores)
//                          a few passes per K-step of the current tile, its residual rows requested half a tile ahead
//                          (own waves because vmcnt is in order per wave: a residual row from HBM must not sit in front
//                          of a staging load the next K-step waits for)
//
// (One consumer, one stager and one finisher wave per SIMD.)
// One workgroup per CU (768 threads, 116.7 KB LDS: two staging stages + the C tile), a static round-robin of tiles per
// workgroup in the XCD-aware order of the other kernels, one barrier per K-step as before. Same K-step order, same six
// products in the same order, same split: the results are the bits of igemm_split_kernel.
// Scope: GEMM-type launches (1x1 / stride 1 / no padding, one geometry segment, no second K segment, no ReLU-adjoint mask),
// 16-byte aligned rows, K >= 64.

constexpr int WS_CLD = 128 + 4;
constexpr size_t WS_LDS_BYTES = (size_t)(2 * 3 * (128 + 128) * SLD) * 4 + (size_t)128 * WS_CLD * 4;

template <int BPRE>
__global__ void __launch_bounds__(768, 1) igemm_ws_kernel(IgemmParams p, int total_tiles) {
  constexpr int BM = 128, BN = 128, CLD = WS_CLD, TM = 2, TN = 2;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  unsigned* As = (unsigned*)smem;                // [2][3][BM][SLD]
  unsigned* Bs = As + 2 * 3 * BM * SLD;          // [2][3][BN][SLD]
  float* Cs = (float*)(Bs + 2 * 3 * BN * SLD);   // [BM][CLD]
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int nk = (p.K + SBK - 1) / SBK;
  const int G = (int)gridDim.x;
  const int T = (total_tiles - (int)blockIdx.x + G - 1) / G;  // tiles of this workgroup (>= 1: grid <= total_tiles)
  const int S = T * nk;                                        // its K-steps, numbered through all its tiles
  const int S4 = (S + 3) & ~3;                                 // every role runs S4 slots (a whole number of ring turns)
  const int tiles_mn = p.tiles_m * p.tiles_n;
  auto tile_of = [&](int j, int& zb, int& m0, int& n0) {
    const int vid = xcd_remap((int)blockIdx.x + j * G, total_tiles);
    zb = vid / tiles_mn;
    const int t = vid - zb * tiles_mn;
    m0 = (t / p.tiles_n) * BM;
    n0 = (t % p.tiles_n) * BN;
  };

  if (wave < 4) {
    // ------------------------------------------------------------------------------------------------ consumers
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int rcol = (lh ^ swz(li)) * 4;
    f32x16 acc[TM][TN];
    u32x4 fa0[3][TM], fb0[3][TN], fa1[3][TM], fb1[3][TN];  // this step's / the next step's fragments (three planes each)
    constexpr int PA[6] = {0, 0, 1, 1, 0, 2};
    constexpr int PB[6] = {0, 1, 0, 1, 2, 0};
    int kc = 0;  // K-step inside the current tile
    // slot s: MFMAs of step s on (fa, fb); fragments of step s + 1 into (na, nb) from LDS[(s + 1) & 1]
    auto slot = [&](int s, const u32x4(&fa)[3][TM], const u32x4(&fb)[3][TN], u32x4(&na)[3][TM], u32x4(&nb)[3][TN]) {
      if (s >= S) {  // (uniform) a pad slot behind the workgroup's last K-step: the barrier only
        __syncthreads();
        return;
      }
      if (kc == 0) {  // (uniform) a tile's first step starts from zero accumulators
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
      }
      const int nb_ = (s + 1) & 1;
      const unsigned* as = As + nb_ * 3 * BM * SLD + (wm * (BM / 2) + li) * SLD + rcol;
      const unsigned* bs = Bs + nb_ * 3 * BN * SLD + (wn * (BN / 2) + li) * SLD + rcol;
      static_for<0, 24>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        constexpr int pq = q / (TM * TN), ti = (q / TN) % TM, tj = q % TN;
        acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[PA[pq]][ti]),
                                                              __builtin_bit_cast(bf16x8, fb[PB[pq]][tj]), acc[ti][tj], 0, 0, 0);
        // (behind the workgroup's last K-step these read a stage nobody multiplies: no branch in the MFMA stream -- a
        // conditional read here split the stream into blocks and cost spills and a wait behind every read)
        if constexpr (q % 2 == 1 && q / 2 < 12) {  // one of the next step's twelve fragment reads behind every second MFMA
          constexpr int r = q / 2, pc = r / (TM + TN), x = r % (TM + TN);
          if constexpr (x < TM) na[pc][x < TM ? x : 0] = *(const u32x4*)(as + (pc * BM + x * 32) * SLD);
          else nb[pc][x < TM ? 0 : x - TM] = *(const u32x4*)(bs + (pc * BN + (x - TM) * 32) * SLD);
        }
      });
      if (++kc == nk) {  // the tile's last step: accumulators -> LDS C tile (the finishers take it from there)
        kc = 0;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            float* cw = Cs + (wm * (BM / 2) + i * 32 + 4 * lh) * CLD + wn * (BN / 2) + j * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) cw[((r & 3) + 8 * (r >> 2)) * CLD] = acc[i][j][r];
          }
      }
      __syncthreads();
    };
    __syncthreads();           // barrier(-1): step 0 is staged
    {
      const unsigned* as = As + (wm * (BM / 2) + li) * SLD + rcol;
      const unsigned* bs = Bs + (wn * (BN / 2) + li) * SLD + rcol;
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) {
#pragma unroll
        for (int i = 0; i < TM; ++i) fa0[pc][i] = *(const u32x4*)(as + (pc * BM + i * 32) * SLD);
#pragma unroll
        for (int j = 0; j < TN; ++j) fb0[pc][j] = *(const u32x4*)(bs + (pc * BN + j * 32) * SLD);
      }
    }
    __syncthreads();           // barrier(0): step 1 is staged
    for (int s = 0; s < S4; s += 2) {
      slot(s, fa0, fb0, fa1, fb1);
      slot(s + 1, fa1, fb1, fa0, fb0);
    }
    return;
  }

  if (wave < 8) {
    // ------------------------------------------------------------------------------------------------ stagers
    const int st = tid - 256;                     // 0..255
    const int c4 = st & 3, r0 = st >> 2;           // float4 c4 of rows r0 + 64 i
    const int wcol = ((c4 >> 1) ^ swz(r0)) * 4 + (c4 & 1) * 2;
    constexpr int NA = 2;                          // fp32 float4s of A per thread and K-step
    constexpr int NB = BPRE ? 3 : 2;               // 16-byte chunks of pre-split B / fp32 float4s of B
    int b_lds[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      if constexpr (BPRE) {
        const int q = st + 256 * j;                // chunk (plane, row, half) of the 3 x 128 x 2
        const int pl = q / (2 * BN), rem = q - pl * (2 * BN);
        const int row = rem >> 1, half = rem & 1;
        b_lds[j] = (pl * BN + row) * SLD + (half ^ swz(row)) * 4;
      } else {
        b_lds[j] = 0;
      }
    }
    // load cursor: the tile and K-step the NEXT load belongs to
    int ld_j = 0, ld_k = 0, zb = 0, m0 = 0, n0 = 0;
    __amdgpu_buffer_rsrc_t ra_src = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0, 0x00020000);
    __amdgpu_buffer_rsrc_t rb_src = ra_src;
    unsigned a_off[NA], b_off[NB];
    const unsigned B_STEP = BPRE ? (unsigned)p.N * 3u * SBK * 2u : SBK * 4;
    auto set_tile = [&]() {
      tile_of(ld_j, zb, m0, n0);
      ra_src = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + (long)zb * p.batch_a), 0, (int)p.a_bytes, 0x00020000);
      rb_src = __builtin_amdgcn_make_buffer_rsrc((void*)(p.Bw + (long)zb * p.batch_b), 0, (int)p.b_bytes, 0x00020000);
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const int m = m0 + r0 + 64 * i;
        a_off[i] = m < p.M ? (unsigned)((m * p.lda + c4 * 4) * 4) : OOB;
      }
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        if constexpr (BPRE) {
          const int q = st + 256 * j;
          const int pl = q / (2 * BN), rem = q - pl * (2 * BN);
          const int n = n0 + (rem >> 1);
          b_off[j] = n < p.N ? (unsigned)((((long)pl * p.N + n) * SBK + (rem & 1) * 8) * 2) : OOB;
        } else {
          const int n = n0 + r0 + 64 * j;
          b_off[j] = n < p.N ? (unsigned)((n * p.ldb + c4 * 4) * 4) : OOB;
        }
      }
    };
    float4 a0[NA], b0[NB], a1[NA], b1[NB], a2[NA], b2[NB], a3[NA], b3[NB];  // a ring of four K-steps in flight
    // The next K-step of the stream. ALWAYS the same number of loads (past the workgroup's last K-step they go to an
    // out-of-range offset and return zeros) and no branch around them: the compiler's vmcnt bookkeeping stays exact, so
    // the staging writes wait for THEIR ring slot only and four K-steps of loads really stay in flight (with a
    // conditional load the waits fell back to vmcnt(0) and every slot paid a whole memory round trip).
    auto load = [&](float4(&ra)[NA], float4(&rb)[NB]) {
      const bool live = ld_j < T;
      if (ld_k == 0 && live) set_tile();  // (scalar / ALU work only)
      const int k0 = ld_k * SBK;
      const bool kin = live && k0 + c4 * 4 < p.K;  // (K % 4 == 0: a float4 is inside K or outside)
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        unsigned off = (b_off[j] != OOB && (BPRE ? live : kin)) ? b_off[j] + (unsigned)ld_k * B_STEP : OOB;
        rb[j] = ldg_b128(rb_src, off);
      }
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        unsigned off = (a_off[i] != OOB && kin) ? a_off[i] + (unsigned)k0 * 4u : OOB;
        ra[i] = ldg_b128(ra_src, off);
      }
      if (++ld_k == nk) {
        ld_k = 0;
        ++ld_j;
      }
    };
    auto store = [&](int buf, const float4(&ra)[NA], const float4(&rb)[NB]) {
      unsigned* as = As + buf * 3 * BM * SLD + r0 * SLD + wcol;
      unsigned* bs = Bs + buf * 3 * BN * SLD + r0 * SLD + wcol;
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        uint2 h, m, l;
        split3(ra[i], h, m, l);
        *(uint2*)(as + (0 * BM + 64 * i) * SLD) = h;
        *(uint2*)(as + (1 * BM + 64 * i) * SLD) = m;
        *(uint2*)(as + (2 * BM + 64 * i) * SLD) = l;
      }
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        if constexpr (BPRE) {
          *(float4*)(Bs + buf * 3 * BN * SLD + b_lds[j]) = rb[j];
        } else {
          uint2 h, m, l;
          split3(rb[j], h, m, l);
          *(uint2*)(bs + (0 * BN + 64 * j) * SLD) = h;
          *(uint2*)(bs + (1 * BN + 64 * j) * SLD) = m;
          *(uint2*)(bs + (2 * BN + 64 * j) * SLD) = l;
        }
      }
    };
    // A K-step's rows are requested FOUR slots before the consumers multiply them (three before their split): one slot is
    // 768 cycles of MFMA = 0.4 us, an L2 hit takes longer than that and an HBM row 1-2 us -- with one slot of lead (the
    // split kernel's distance, where two interleaved workgroups double the slot) the slot WAS the load latency.
    //   slot s: split + staging writes of step s + 2 from ring[(s + 2) % 4], then loads of step s + 6 into the same ring slot
    load(a0, b0);              // step 0
    load(a1, b1);              // step 1
    load(a2, b2);              // step 2
    load(a3, b3);              // step 3
    store(0, a0, b0);          // step 0
    __syncthreads();           // barrier(-1)
    load(a0, b0);              // step 4
    store(1, a1, b1);          // step 1
    load(a1, b1);              // step 5
    __syncthreads();           // barrier(0)
    // The staging writes go FIRST in a slot (their rows landed slots ago; the barrier at the slot's end waits for them), the
    // requests for the ring slot they free behind them. (s_setprio 1 for the helper waves measured +-0.)
    for (int s = 0; s < S4; s += 4) {  // (writes behind the last K-step stage zeros nobody reads)
      store(0, a2, b2);        // step s + 2
      load(a2, b2);            // step s + 6
      __syncthreads();
      store(1, a3, b3);        // step s + 3
      load(a3, b3);            // step s + 7
      __syncthreads();
      store(0, a0, b0);        // step s + 4
      load(a0, b0);            // step s + 8
      __syncthreads();
      store(1, a1, b1);        // step s + 5
      load(a1, b1);            // step s + 9
      __syncthreads();
    }
    return;
  }

  // -------------------------------------------------------------------------------------------------- finishers
  {
    const int et = tid - 512;                      // 0..255
    const int ec = (et & 31) * 4, er = et >> 5;     // float4 ec of rows er + 8 q, q = 0..15
    constexpr int NP = 16, NH = 8;
    const int h = nk / 2;                           // passes 0..7 in slots [0, h), passes 8..15 in slots [h, nk - 1)
    const int pps0 = (NH + h - 1) / h, pps1 = (NH + (nk - 1 - h) - 1) / (nk - 1 - h);
    float4 rA[NH], rB[NH];                          // residual rows of the passes, requested half a tile ahead
    float sc[4], sh[4];
    int ep_zb = 0, ep_m0 = 0, ep_n0 = 0;            // the tile being finished
    int cur_zb = 0, cur_m0 = 0, cur_n0 = 0;         // the tile being computed (its residual rows are requested)
    const bool has_res = p.residual != nullptr;
    auto req = [&](float4(&r)[NH], int q0, int zb, int m0, int n0) {
      if (!has_res) return;
#pragma unroll
      for (int q = 0; q < NH; ++q) {
        const int m = m0 + er + 8 * (q0 + q), n = n0 + ec;
        r[q] = (m < p.M && n < p.N) ? *(const float4*)(p.residual + (long)m * p.ldr + n) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    };
    auto begin_tile = [&]() {  // the computed tile becomes the one to finish: its column constants
      ep_zb = cur_zb;
      ep_m0 = cur_m0;
      ep_n0 = cur_n0;
      const int n = ep_n0 + ec;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const bool nok = n + t < p.N;
        sc[t] = ((nok && p.scale) ? p.scale[n + t] : 1.f) * p.alpha;
        sh[t] = (nok && p.shift) ? p.shift[n + t] : 0.f;
      }
    };
    auto pass = [&](int q, const float4& res) {
      const int rr = er + 8 * q, m = ep_m0 + rr, n = ep_n0 + ec;
      if (m >= p.M || n >= p.N) return;
      const float4 a4 = *(const float4*)(Cs + rr * CLD + ec);
      float v[4] = {a4.x * sc[0] + sh[0], a4.y * sc[1] + sh[1], a4.z * sc[2] + sh[2], a4.w * sc[3] + sh[3]};
      if (has_res) {
        v[0] += res.x;
        v[1] += res.y;
        v[2] += res.z;
        v[3] += res.w;
      }
      if (p.relu) {
#pragma unroll
        for (int t = 0; t < 4; ++t) v[t] = fmaxf(v[t], 0.f);
      }
      *(float4*)(p.C + (long)ep_zb * p.batch_c + (long)m * p.ldc + n) = make_float4(v[0], v[1], v[2], v[3]);
    };
    auto passes = [&](int q_lo, int q_hi) {  // (q bounds are uniform; the arrays are indexed by constants)
#pragma unroll
      for (int q = 0; q < NP; ++q)
        if (q >= q_lo && q < q_hi) pass(q, q < NH ? rA[q < NH ? q : 0] : rB[q < NH ? 0 : q - NH]);
    };
    __syncthreads();  // barrier(-1)
    __syncthreads();  // barrier(0)
    int kc = 0, jt = 0;
    int ep_next = NP;  // next pass of the tile that waits in the C tile (NP: none / done)
    tile_of(0, cur_zb, cur_m0, cur_n0);
    for (int s = 0; s < S4; ++s) {
      if (ep_next < NP && kc < nk - 1) {  // (never in the slot in which the consumers dump the next tile)
        const int hi = kc < h ? min(NH, (kc + 1) * pps0) : min(NP, NH + (kc - h + 1) * pps1);
        if (hi > ep_next) {
          passes(ep_next, hi);
          ep_next = hi;
        }
      }
      if (jt < T) {
        if (kc == h) req(rA, 0, cur_zb, cur_m0, cur_n0);          // (set A is free: its passes ran in slots [0, h))
        if (kc == nk - 1) req(rB, NH, cur_zb, cur_m0, cur_n0);    // (set B is free: its passes ended in slot nk - 2)
      }
      __syncthreads();
      if (++kc == nk) {
        kc = 0;
        if (jt < T) {  // the consumers dumped tile jt into the C tile during this slot
          begin_tile();
          ep_next = 0;
          if (++jt < T) tile_of(jt, cur_zb, cur_m0, cur_n0);
        }
      }
    }
    passes(ep_next, NP);  // what is left of the workgroup's last tile
  }
}
