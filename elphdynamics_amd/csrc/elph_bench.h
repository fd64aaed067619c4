// elph_bench.h — PRIVATE measurement hooks of libelphgpu (bench.py, tools/): exported by the library, NOT part of the drop-in ABI
// (include/elph_gpu.h).  Nothing a caller of the operator API needs; signatures may change between rounds.
#pragma once
#include "../../include/elph_gpu.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Measurement of one hot-path unit with inputs resident in HBM (no host traffic in the timed region).
 * what: 0 = MᵀM apply, 1 = one un-preconditioned CG iteration of the two-kernel (streaming) form (k_cg_ap + k_cg_xr, stop
 *       test disabled), 2 = KPM apply, 3 = one preconditioned CG iteration, 4 = k_cg_ap alone, 5 = k_cg_xr alone,
 *       6 / 7 / 8 = the forward transform / Chebyshev recursion / inverse transform of the KPM apply alone (as the
 *       preconditioned iteration launches them), 9 = `reps` un-preconditioned CG iterations of every right-hand side in ONE
 *       launch of the workgroup-resident kernel (cg_wg.hip: the form elph_ldiv/elph_cg_solve use when elph_bench_wg_info
 *       says it applies; needs a fresh elph_bench_prepare before every run), 10 = `reps` KPM-PRECONDITIONED iterations in one
 *       launch of the resident preconditioned kernel (pcg_wg.hip: what elph_ldiv with a preconditioner runs for 1..8 right-hand
 *       sides on the 16 x 16 square lattice; ELPH_E_UNSUPPORTED elsewhere; fresh elph_bench_prepare(…, 10, …) before every run), 11 = `reps` preconditioned iterations
 *       of the batch as TWO half-batches on two streams (the form elph_ldiv_batched runs from 192 right-hand sides; prepare with what = 3;
 *       ELPH_E_UNSUPPORTED where the halves are not whole groups of chains or the p/x-fused iteration does not apply),
 *       12 = `reps` un-preconditioned iterations of every right-hand side in the SLAB form of a lattice beyond 320 sites (slabs.hip: the resident
 *       kernel on slabs of rows, all on this device, one launch per right-hand side; prepare with what = 1; *ms_total = sum of the launches).
 * elph_bench_prepare: loads nrhs right-hand sides (B: host, reference layout, nrhs*ndim; NULL keeps what the
 *   last solve left on the device), zeroes x, seeds the CG state with tol = 0 (never converges).
 * elph_bench_run: launches `reps` units back-to-back on the handle's stream (captured graph chunks when
 *   use_graph != 0 and reps is a multiple of the chunk), brackets them with HIP events recorded on that
 *   stream, synchronises, and returns the event time in ms (total, not per rep). */
int elph_bench_prepare(elph_handle h, int what, int nrhs, const double *B);
int elph_bench_run(elph_handle h, int what, int nrhs, int reps, int use_graph, double *ms_total);
/* Which k_cg_ap variant a batch of nrhs uses: *slices_per_wave = 1 (k_cg_ap_fast / generic) or T (k_cg_ap_chunk<T>). */
int elph_bench_info(elph_handle h, int nrhs, int *slices_per_wave);
/* Whether un-preconditioned solves of a batch of nrhs right-hand sides run as the workgroup-resident kernel (*usable = 1) and
 * its shape: T tau-slices per wavefront, W wavefronts per workgroup, G workgroups per right-hand side. */
int elph_bench_wg_info(elph_handle h, int nrhs, int *usable, int *T, int *W, int *G);

/* Whether the LAST preconditioned solve / elph_bench_prepare ran its iteration p/x-fused (*fused = 1: x += alpha p and p = P^-1 r + beta p
 * in the epilogue of the inverse tau-transform, k_cg_ap_chunk<PX> reading the ready p; kernels.hip: px_plan). */
int elph_bench_px_info(elph_handle h, int *fused);

/* The patch layout of this handle (pgrid.hip), if any: *kind 0 none, 1 square, 2 honeycomb, 3 triangular; the patch shape and the wavefronts per time
 * slice; *tables = 1 when the hopping is disordered and the mat-vec / Chebyshev kernels take the patch layout with a (cosh, sinh) table in LDS. */
int elph_bench_pg_info(elph_handle h, int *kind, int *px, int *py, int *nw, int *tables);

/* Whether an un-preconditioned solve of nrhs right-hand sides FROM x = 0 on this handle runs in the slab form (slabs.hip: lattices beyond
 * 320 sites as slabs of rows on the same device, the resident kernel per slab, one launch) and its shape. */
int elph_bench_slabs_info(elph_handle h, int nrhs, int *usable, int *slabs, int *sites_per_slab, int *own_sites);

/* the sharded solve (shard.hip): exactly `iters` iterations (no stop test); *ms = HIP-event time of this rank's launch.  b_slab may
 * be NULL (keeps the right-hand side of the previous call).  Needs elph_shard_prepare + barrier like a solve. */
int elph_shard_iterate(elph_handle h, const double *b_slab, int64_t iters, double *ms);

#ifdef __cplusplus
}
#endif
