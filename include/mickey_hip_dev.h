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

/* Kernel variant of mk_flash_attn_fwd: 0 = automatic (2 for >= 512 workgroups of 256 queries, 1 for smaller grids),
 * 1 / 2 = the production kernel with 32 / 64 queries per wave (lean softmax: running maximum folded into the MFMA
 * accumulator init, re-based only when a tile outgrows it, seen in the row sums; fp32 row sums on the VALU), 3 = the classic online-softmax
 * kernel (64 queries per wave; A/B partner).  Process-wide; for benchmarks and tests. */
int mk_attn_set_mode(int mode);

/* mk_dual_softmax_split, pass 2 (the writer of scores / kp_scores / final_scores): column chunks per 32-row block = waves
 * that share one block's rows; 0 = one pair of column tiles per wave (default), 8 = round 4's first version (A/B). */
int mk_dual_softmax_set_chunks(int chunks);

/* mk_sinkhorn: n > 0 = n image pairs iterated together through all Sinkhorn iterations (so that their coupling matrices could
 * stay in the 256-MB Infinity Cache between the 20 passes; measured no faster: profiles/r04c_bench_matcher.txt), 0 = the
 * whole batch per pass with non-temporal reads (default). */
int mk_sinkhorn_set_group(int pairs);

#ifdef __cplusplus
}
#endif
#endif /* MICKEY_HIP_DEV_H */
