// cg_fast.hip — latency-tuned gfx950 kernels for lattices whose checkerboard has <= 6 colours
// (every even-L square, honeycomb and triangular lattice of the reference's example decks).
// The kernels live in cg_fast_impl.inc, compiled per lane-program width — 4 colours (namespace lp4) and 6 (lp6) — and per sites-per-lane
// count (cg_fast_npl.hip, 8 translation units each; this file holds what is common to the counts: cg_fast_shared.inc) —
// the colour count is a compile-time loop bound (bonds live in registers, the fused forward/reverse sweep pairs colour
// cc with colour MC-1-cc), so a 4-colour lattice does not pay for the two stages only triangular lattices need.
//
// What differs from the generic kernels in kernels.hip (same arithmetic, same results):
//   * "lane program": the bond list is re-packed on the host per colour into [colour][pass][lane]
//     so that every lane keeps ITS bonds (site pair + cosh + sinh) in registers for the whole kernel;
//     a checkerboard colour is then  LDS read -> 4 FMAs -> LDS write -> wave barrier  with no
//     dependent global load inside the sweep (the generic kernel pays one L2 round trip per colour);
//   * every global load of the kernel (state, partial sums, vectors, exp(-dtau V), lane program)
//     is independent of every other and issued up front: one memory round trip per kernel;
//   * the p ping-pong index is a launch constant (launch parity) instead of device state;
//   * XCD-aware 1-D grid: workgroups with equal (blockIdx.x % 8) share an XCD/L2, so they are given
//     consecutive tau-slices — the tau+-1 halo re-reads of the fused MtM then hit the same L2.
//
// Reference semantics: see kernels.hip / SURVEY.md Appendix A.

#include "cg_fast_common.h"
#include "kpm_sq_dev.h"

#define ELPH_LP_MC 4
#define LPNS lp4
#include "cg_fast_shared.inc"
#undef ELPH_LP_MC
#undef LPNS

// the 6-colour instantiation lives in cg_fast6.hip
namespace lp6 {
int elph_fast_mul(elph_handle_s *h, int which, double *yS, const double *vS, int nvec);
int elph_fast_cg_ap(elph_handle_s *h, const CgBufs &B, int nrhs, int parity, bool px);
int elph_fast_cg_xr(elph_handle_s *h, const CgBufs &B, int nrhs, int parity);
int elph_fast_kpm_cheb(elph_handle_s *h, int nrhs, const CgState *st, double *rz_part, int nrz, bool *did_rz, const double *rr_part, int fold_nct);
}  // namespace lp6
// ---- dispatch on the handle's lane-program width ------------------------------------------------------------
int elph_fast_mul(elph_handle_s *h, int which, double *yS, const double *vS, int nvec) {
    return h->lp_mc == 4 ? lp4::elph_fast_mul(h, which, yS, vS, nvec) : lp6::elph_fast_mul(h, which, yS, vS, nvec);
}
int elph_choose_T(const elph_handle_s *h, int nrhs) { return lp4::elph_choose_T(h, nrhs); }
int elph_choose_T_px(const elph_handle_s *h, int nrhs) { return lp4::elph_choose_T_px(h, nrhs); }
int elph_fast_cg_ap(elph_handle_s *h, const CgBufs &B, int nrhs, int parity, bool px) {
    // the p/x-fused step of a preconditioned batch on the 16 x 16 square lattice: the checkerboard in registers (cg_sq16.hip)
    h->sq16_ap_ran = px && B.npap > 0 && B.npap < (int)h->L && (int)h->L % B.npap == 0 && elph_sq16_ap_usable(h, (int)h->L / B.npap);
    if (h->sq16_ap_ran) return elph_sq16_cg_ap_px(h, B, nrhs, parity);
    return h->lp_mc == 4 ? lp4::elph_fast_cg_ap(h, B, nrhs, parity, px) : lp6::elph_fast_cg_ap(h, B, nrhs, parity, px);
}
int elph_fast_cg_xr(elph_handle_s *h, const CgBufs &B, int nrhs, int parity) {
    return h->lp_mc == 4 ? lp4::elph_fast_cg_xr(h, B, nrhs, parity) : lp6::elph_fast_cg_xr(h, B, nrhs, parity);
}
int elph_fast_kpm_cheb(elph_handle_s *h, int nrhs, const CgState *st, double *rz_part, int nrz, bool *did_rz, const double *rr_part, int fold_nct) {
    return h->lp_mc == 4 ? lp4::elph_fast_kpm_cheb(h, nrhs, st, rz_part, nrz, did_rz, rr_part, fold_nct)
                         : lp6::elph_fast_kpm_cheb(h, nrhs, st, rz_part, nrz, did_rz, rr_part, fold_nct);
}

