// cg_fast.hip — latency-tuned gfx950 kernels for lattices whose checkerboard has <= 6 colours
// (every even-L square, honeycomb and triangular lattice of the reference's example decks).
// The kernels live in cg_fast_impl.inc, compiled twice: lane programs of 4 colours (namespace lp4) and of 6 (lp6) —
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

#include <cstdlib>

#include "elph_internal.h"

#define WAVE ELPH_WAVE

// Ordering of LDS traffic inside ONE wavefront.  Every slab in this file is private to a wave, and the LDS
// pipeline executes a wave's DS instructions in issue order, so a ds_read issued after a ds_write of the same
// wave observes it without any s_waitcnt/s_barrier in between.  All that is needed is that the COMPILER keeps
// the program order of possibly-aliasing LDS accesses: a pure compiler barrier, no instruction.
// (Using __syncthreads() here costs an s_waitcnt lgkmcnt(0) per colour: one extra LDS round trip per stage.)
#ifdef ELPH_LDS_SYNC
#define WAVE_LDS_ORDER() __syncthreads()
#else
#define WAVE_LDS_ORDER() asm volatile("" ::: "memory")
#endif

// device-coherent scalar traffic (experiment): agent-scope relaxed atomics => sc1 loads/stores that bypass L1/K$
__device__ __forceinline__ double ld_coh(const double *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ CgState ld_state(const CgState *p) {
    CgState s;
    const unsigned long long *q = reinterpret_cast<const unsigned long long *>(p);
    unsigned long long w[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) w[i] = __hip_atomic_load(q + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_memcpy(&s, w, sizeof(CgState));
    return s;
}

__device__ __forceinline__ double wave_sum2(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}

__device__ __forceinline__ double reduce_partials2(const double *p, int n) {
    double a = 0.0;
    for (int i = threadIdx.x; i < n; i += WAVE) a += ld_coh(p + i);
    return wave_sum2(a);
}

// XCD-aware mapping of a 1-D grid of 8*C*nrhs workgroups onto (tau, rhs); C = ceil(L/8)
__device__ __forceinline__ bool xcd_map(int L, int &t, int &rhs) {
    const int b = blockIdx.x;
    const int C = (L + 7) >> 3;
    const int xcd = b & 7, k = b >> 3;
    rhs = k / C;
    t = xcd * C + (k - rhs * C);
    return t < L;
}


#define ELPH_LP_MC 4
#define LPNS lp4
#include "cg_fast_impl.inc"
#undef ELPH_LP_MC
#undef LPNS
#define ELPH_LP_MC 6
#define LPNS lp6
#include "cg_fast_impl.inc"
#undef ELPH_LP_MC
#undef LPNS

// ---- dispatch on the handle's lane-program width ------------------------------------------------------------
int elph_fast_mul(elph_handle_s *h, int which, double *yS, const double *vS, int nvec) {
    return h->lp_mc == 4 ? lp4::elph_fast_mul(h, which, yS, vS, nvec) : lp6::elph_fast_mul(h, which, yS, vS, nvec);
}
int elph_choose_T(const elph_handle_s *h, int nrhs) { return lp4::elph_choose_T(h, nrhs); }
int elph_fast_cg_ap(elph_handle_s *h, const CgBufs &B, int nrhs, int parity) {
    return h->lp_mc == 4 ? lp4::elph_fast_cg_ap(h, B, nrhs, parity) : lp6::elph_fast_cg_ap(h, B, nrhs, parity);
}
int elph_fast_cg_xr(elph_handle_s *h, const CgBufs &B, int nrhs, int parity) {
    return h->lp_mc == 4 ? lp4::elph_fast_cg_xr(h, B, nrhs, parity) : lp6::elph_fast_cg_xr(h, B, nrhs, parity);
}
int elph_fast_kpm_cheb(elph_handle_s *h, int nrhs, const CgState *st, double *rz_part, int nrz, bool *did_rz, const double *rr_part) {
    return h->lp_mc == 4 ? lp4::elph_fast_kpm_cheb(h, nrhs, st, rz_part, nrz, did_rz, rr_part)
                         : lp6::elph_fast_kpm_cheb(h, nrhs, st, rz_part, nrz, did_rz, rr_part);
}
int elph_fast_cg_resident(elph_handle_s *h, const CgBufs &B, int nrhs, bool *ran) {
    if (h->lp_mc != 4) { *ran = false; return ELPH_OK; }
    return lp4::elph_fast_cg_resident(h, B, nrhs, ran);
}
