/* mickey_hip_dev.h -- development knobs of libmickey_hip.so (benchmarks, tests, A/B tools).
 *
 * NOT part of the drop-in ABI (include/mickey_hip.h): these set PROCESS-WIDE state that selects between kernel
 * schedules computing the same result.  The product path never calls them.
 */
#ifndef MICKEY_HIP_DEV_H
#define MICKEY_HIP_DEV_H

#ifdef __cplusplus
extern "C" {
#endif

/* Schedule of the GEMM / implicit-GEMM conv kernel: 0 = automatic (128x128 tiles for small problems, the default
 * 256x256 schedule for large ones; 64x128 tiles where 128x128 would leave CUs idle), 1 = force 128x128 (4 waves, 2
 * workgroups per CU), 2 = force 64x128 (same kernel, half the rows, 3 workgroups per CU), 7 = force the 8-wave ping-pong
 * 256x256 schedule (two waves per SIMD);
 * 400 + b sets the band height b (in m-tiles) of the 256x256 tile order; 500 / 501 switch the automatic use of the 64x128
 * tiling off / on. */
int mk_gemm_set_tile(int mode);

/* Kernel variant of mk_flash_attn_fwd: 0 = automatic (the large-grid kernel for >= 512 workgroups of 256 queries, 4 for
 * small grids), 1 = 32 queries/wave, 2 = 64 queries/wave (both two waves per SIMD), 4 = VALU-lean (max folded into the
 * MFMA accumulator init, row sums on the matrix pipe), 7 = one wave per SIMD (64 queries per wave, K / V^T fragments and
 * the output accumulators in the accumulator register file, hand-placed MFMA / softmax interleave; problems with fewer
 * than 4 KV tiles run variant 2).  Process-wide; for benchmarks and tests. */
int mk_attn_set_mode(int mode);

#ifdef __cplusplus
}
#endif
#endif /* MICKEY_HIP_DEV_H */
