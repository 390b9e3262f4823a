/* libdana_hip.so -- debug / tuning / profiling switches. NOT part of the drop-in ABI (include/dana_hip.h): nothing the
 * reference binds goes through these, no product code path depends on a non-default value, and they may change between
 * rounds. They exist for the A/B tools under tools/ and for the bit-identity tests that compare two forms of one kernel.
 * All are process-wide configuration calls (issue them between forwards, not while other threads are inside one). */
#ifndef DANA_HIP_DEBUG_H
#define DANA_HIP_DEBUG_H
#include "dana_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Force the tile shape of the split kernel (dana_set_mfma_mode(1)) for every launch that can take it: 0 = the dispatcher's
 * own choice (default), 2 = 128x64, 3 = 64x64, 4 = 128x128, 5 = 64x128 (tools/tile_sweep.py). No effect on the f32-MFMA
 * kernel; dana_set_mfma_mode() resets it. */
int dana_debug_force_tile(int tile);
/* Epilogue form of the split kernel. 0 (default): scale / shift / residual / ReLU / ReLU-adjoint mask run on the
 * accumulator registers and the results leave as dword buffer stores (a 32x32 accumulator row = 32 consecutive channels =
 * one 128-byte segment per row and half-wave): no LDS C tile, 49 instead of 67.6 KB of LDS per 128x128 tile. 1: the
 * round-1..4 form through an LDS C tile (float4 rows). Same arithmetic in the same order -> the same bits
 * (tests/test_gpu_contractions.py uses mode 1 as the reference of test_register_epilogue_*). */
int dana_set_epilogue_mode(int mode);
int dana_get_epilogue_mode(void);
/* While `buffer` is non-null every split-kernel block writes eight 64-bit words {shader-clock at start, at the first
 * K-step, after the K loop, at the end, HW_ID, 100 MHz wall clock at the end, wall clock at the start, 0} at
 * buffer[(z * grid + block) * 8]. The pointer is read when a launch is ISSUED, so two launches issued with two buffers can
 * run concurrently (tools/overlap_probe.py). The caller sizes the buffer for the launches it traces; null switches it
 * off. (tools/igemm_trace.py, gemm_power.py) */
int dana_set_igemm_trace(unsigned long long* buffer);
/* dana_topk_desc / dana_sort_desc dispatch: 0 = the measured rule (default), 1 = the sample sort for every row, 2 = the
 * single-workgroup kernel wherever it can run (n <= 40 960, min(topn, n) <= 12 288). */
int dana_set_sort_mode(int mode);

/* A HIP stream restricted to a set of compute units (hipExtStreamCreateWithCUMask), for the overlap experiments of
 * profiles/r6_overlap.md. mask_words 32-bit words, bit i of the mask = CU (i / n_xcd) of XCD (i % n_xcd) on a multi-XCD part
 * in SPX mode (the KFD deals the bits round-robin over the XCDs), 256 bits on an MI355X. EVERY XCD must keep at least one
 * CU: the command processor deals a launch's workgroups over all XCDs whatever the mask says, and an XCD without CUs
 * never retires its share (the call refuses such masks). *stream_out receives the hipStream_t. */
int dana_debug_stream_create_cumask(const unsigned int* mask, int mask_words, int n_xcd, dana_stream_t* stream_out);
int dana_debug_stream_destroy(dana_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
