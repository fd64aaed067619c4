// kpm_sq_dev.h — the register-exchange Chebyshev recursions (even-L square lattices; honeycomb of 12 x 12 cells) (KPMPreconditioners.jl:606-693) as device
// functions: used by k_kpm_cheb_sq (cg_fast_impl.inc, the streaming KPM apply) and by the resident preconditioned solver (pcg_wg.hip).
#pragma once
#include "cg_fast_common.h"

namespace kpmsq {

template <int P>
struct SqLane {
    static constexpr int NS = P * P;
    double c[4][P * P], s[4][P * P];     // per colour, per own site: cosh/sinh of the bond touching it
    double cu, su;                       // UNI: every bond has the same cosh/sinh (no hopping disorder) -> two scalars, 64 VGPRs less
    int xp, xm, yp, ym, xe, ye;          // partner lanes: +x, -x, +y, -y neighbours; P == 1: x-even / y-even partner
};

#define SQC(col, i) (UNI ? T.cu : T.c[col][i])
#define SQS(col, i) (UNI ? T.su : T.s[col][i])
template <int P, bool REVERSE, bool UNI>
__device__ __forceinline__ void sq_cb_apply(double (&v)[P * P], const SqLane<P> &T) {
    // slot index: dx + P*dy
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
        const int col = REVERSE ? 3 - cc : cc;
        if (P == 2) {
            if (col == 0 || col == 2) {          // in-lane pairs: (0,d)-(1,d) along x, (d,0)-(d,1) along y
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    const int i = (col == 0) ? (0 + 2 * d) : (d + 0), j = (col == 0) ? (1 + 2 * d) : (d + 2);
                    const double t0 = v[i], t1 = v[j];
                    v[i] = SQC(col, i) * t0 + SQS(col, i) * t1;
                    v[j] = SQC(col, j) * t1 + SQS(col, j) * t0;
                }
            } else {                             // cross-lane: my high-side sites pair with the +neighbour's low-side sites
                const int up = (col == 1) ? T.xp : T.yp, dn = (col == 1) ? T.xm : T.ym;
                double fromUp[2], fromDn[2];
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    const int lo = (col == 1) ? (0 + 2 * d) : (d + 0), hi = (col == 1) ? (1 + 2 * d) : (d + 2);
                    fromUp[d] = __shfl(v[lo], up, WAVE);      // neighbour's low-side value -> partner of my high-side site
                    fromDn[d] = __shfl(v[hi], dn, WAVE);      // neighbour's high-side value -> partner of my low-side site
                }
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    const int lo = (col == 1) ? (0 + 2 * d) : (d + 0), hi = (col == 1) ? (1 + 2 * d) : (d + 2);
                    v[hi] = SQC(col, hi) * v[hi] + SQS(col, hi) * fromUp[d];
                    v[lo] = SQC(col, lo) * v[lo] + SQS(col, lo) * fromDn[d];
                }
            }
        } else {                                 // P == 1: one site per lane, every colour is a lane permutation
            const int partner = (col == 0) ? T.xe : (col == 2) ? T.ye : (col == 1) ? T.xp : T.yp;   // xp/yp hold the odd-colour partner
            const double t = __shfl(v[0], partner, WAVE);
            v[0] = SQC(col, 0) * v[0] + SQS(col, 0) * t;
        }
    }
}

// The 16 x 16 lattice in the lane layout of the workgroup-resident CG (cg_wg.hip, sq_patch_site): lane l holds the 2 x 2 patch
// X = (l >> 1) & 7, Y = 2 (l >> 4) + (l & 1), rows stored in reverse order for odd Y.  x-even and y-even bonds pair registers of
// one lane; x-odd bonds: four DPP row rotations by 2 lanes; y-odd bonds: registers 2, 3 swap inside the lane pair (DPP quad
// swap), registers 0, 1 cross to the neighbouring group of 16 lanes (one ds_bpermute pair each).  12 DPP moves + 4 ds_bpermute
// per apply (the column-segment layout of round 2: 24 + 4; the 8 x 8 patch layout above: 16 ds_bpermute) — the recursion is a
// chain of ~120 dependent applies at the lowest frequency, one wave issues one vector instruction every 4-5 cycles.
// UNI: a colour is c (I + th P_colour) with th = sinh/cosh; the apply does the bracket (one fma per site and colour) and the
// caller multiplies by c^4 where it scales the result anyway (T.su = th, T.cu = c^4).
// `mid` runs between the issue of the ds_bpermute pairs of the y-odd colour and the use of their results: work that does not
// depend on the apply (the caller's) fills that LDS round trip.  (T.yp = sq_patch_ycross(lane).)
template <bool REVERSE, bool UNI, class F>
__device__ __forceinline__ void sq16_cb_apply(double (&v)[4], const SqLane<2> &T, F &&mid) {
    auto upd = [&T](int col, int k, double x, double t) { return UNI ? x + T.su * t : T.c[col][k] * x + T.s[col][k] * t; };
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
        const int col = REVERSE ? 3 - cc : cc;
        if (col == 0) {
            const double n0 = upd(0, 0, v[0], v[1]), n1 = upd(0, 1, v[1], v[0]), n2 = upd(0, 2, v[2], v[3]), n3 = upd(0, 3, v[3], v[2]);
            v[0] = n0; v[1] = n1; v[2] = n2; v[3] = n3;
        } else if (col == 1) {
            const double t1 = dpp_f64<0x12E>(v[0]), t3 = dpp_f64<0x12E>(v[2]);     // row_ror:14 = lane + 2
            const double t0 = dpp_f64<0x122>(v[1]), t2 = dpp_f64<0x122>(v[3]);     // row_ror:2  = lane - 2
            v[0] = upd(1, 0, v[0], t0); v[1] = upd(1, 1, v[1], t1); v[2] = upd(1, 2, v[2], t2); v[3] = upd(1, 3, v[3], t3);
        } else if (col == 2) {
            const double n0 = upd(2, 0, v[0], v[2]), n2 = upd(2, 2, v[2], v[0]), n1 = upd(2, 1, v[1], v[3]), n3 = upd(2, 3, v[3], v[1]);
            v[0] = n0; v[1] = n1; v[2] = n2; v[3] = n3;
        } else {
            const double c0 = __shfl(v[0], T.yp, WAVE), c1 = __shfl(v[1], T.yp, WAVE);
            const double t2 = dpp_f64<0xB1>(v[2]), t3 = dpp_f64<0xB1>(v[3]);       // quad_perm [1,0,3,2]
            v[2] = upd(3, 2, v[2], t2); v[3] = upd(3, 3, v[3], t3);
            mid();
            v[0] = upd(3, 0, v[0], c0); v[1] = upd(3, 1, v[1], c1);
        }
    }
}
#undef SQC
#undef SQS

// The honeycomb lattice of 12 x 12 cells (config D) in the reference's colouring [A-B of a cell | B(x,y)-A(x+1,y) | B(x,y)-A(x,y+1)]
// (detect_honeycomb12), QUAD layout of the Chebyshev recursion: lane 4 y + i (48 of the 64 lanes) holds the three cells x = 3 i + b of
// lattice row y — six sites, register 2 b + orbital.  12 = 3 cells in a lane x 4 lanes of a DPP quad, so the x-direction wraps where
// quad_perm wraps: A-B pairs registers (0,1), (2,3), (4,5); B(x,y)-A(x+1,y) pairs (1,2), (3,4) of the lane and sends register 5 to
// register 0 of the next lane of the quad (two DPP moves of an f64); B(x,y)-A(x,y+1) crosses to the lane 4 up / 4 down, cyclically
// in 48 (six ds_bpermute pairs, issued together: one LDS round trip, with `mid` inside it).  18 fma + 4 DPP moves + 12 ds_bpermute
// per apply — no LDS slab, no mirror lanes to repair (the resident CG, cg_wg_dev.h, uses mirror lanes instead: there p is re-made
// pointwise every iteration; a recursion has nothing that repairs them).  Uniform hopping: a colour is c (I + th P), T.th = s/c;
// the caller folds c^3 into the scale of A'.
struct HcLane { double th; int up, dn; };       // up / dn: lane + 4 / lane - 4 modulo 48 (idle lanes 48..63: themselves)

using ::hc12q_site;        // (cg_fast_common.h: shared with the resident CG)

template <bool REVERSE, class F>
__device__ __forceinline__ void hc12q_cb_apply(double (&v)[6], const HcLane &T, F &&mid) {
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) {
        const int col = REVERSE ? 2 - cc : cc;
        if (col == 0) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const double a = v[2 * j] + T.th * v[2 * j + 1], b = v[2 * j + 1] + T.th * v[2 * j];
                v[2 * j] = a; v[2 * j + 1] = b;
            }
        } else if (col == 1) {
            const double t0 = dpp_f64<0x93>(v[5]);           // quad_perm [3,0,1,2]: the lane below in the quad
            const double t5 = dpp_f64<0x39>(v[0]);           // quad_perm [1,2,3,0]: the lane above
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const double b = v[2 * j + 1] + T.th * v[2 * j + 2], a = v[2 * j + 2] + T.th * v[2 * j + 1];
                v[2 * j + 1] = b; v[2 * j + 2] = a;
            }
            v[0] += T.th * t0; v[5] += T.th * t5;
        } else {
            double fa[3], fb[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) { fa[j] = __shfl(v[2 * j + 1], T.dn, WAVE); fb[j] = __shfl(v[2 * j], T.up, WAVE); }
            mid();
#pragma unroll
            for (int j = 0; j < 3; ++j) { v[2 * j] += T.th * fa[j]; v[2 * j + 1] += T.th * fb[j]; }
        }
    }
}

// u_1 = v, u_2 = A'u_1, u_{n+1} = 2 A'u_n - u_{n-1} with A' = a A - b (mulA'!, KPMPreconditioners.jl:685-693; A = CB diag(eb),
// transposed: diag(eb) CB^T) on NS values per lane; apply(w, mid) is the checkerboard of the lattice's lane layout (in place; it runs
// mid() where a cross-lane round trip leaves room).  The scale a (and the 2 of the recurrence) ride on the diagonal the step
// multiplies by anyway: e1 = a eb, e2 = 2 a eb, so a step is the checkerboard apply plus 4 instructions per site (one fewer than
// scaling, A', 2 A'u - u separately), and the two history vectors swap roles instead of being copied — the recursion is a dependent
// chain of up to 2 (order - 1) of these steps, and one wave issues one vector instruction every 4-5 cycles.
template <int NS, bool TRANSPOSED, class APPLY>
__device__ __forceinline__ void kpm_series(double (&Pacc)[NS], double (&Qacc)[NS], const double (&vin)[NS], const double (&eb)[NS],
                                           const double2 *c, int order, double a, double b, APPLY &&apply) {
    double ua[NS], ub[NS], e1[NS], e2[NS];
    const double b2 = 2.0 * b;
    {
        const double2 c0 = c[0];
#pragma unroll
        for (int q = 0; q < NS; ++q) { Pacc[q] = c0.x * vin[q]; Qacc[q] = c0.y * vin[q]; ua[q] = vin[q]; ub[q] = 0.0; e1[q] = a * eb[q]; e2[q] = 2.0 * e1[q]; }
    }
    // one step: un = the latest vector, um = the one before (overwritten by the new one)
    // The coefficient sums of a step's result (P += Re c_n u_n, Q += Im c_n u_n) and the history term of the next step do not
    // depend on the next checkerboard apply: they run inside it, while its ds_bpermute pair is in flight (the apply's `mid`) —
    // ~12 of the 60 instructions of a step off the dependent chain.
    double2 cpend = make_double2(0.0, 0.0);               // coefficient of the vector in `un` whose sums are still pending
    bool pend = false;
    auto step = [&](double (&un)[NS], double (&um)[NS], const double (&e)[NS], double bb, bool first, const double2 cn) {
        double w[NS], t[NS];
#pragma unroll
        for (int q = 0; q < NS; ++q) w[q] = TRANSPOSED ? un[q] : e[q] * un[q];
        auto mid = [&]() {
#pragma unroll
            for (int q = 0; q < NS; ++q) {
                t[q] = first ? bb * un[q] : bb * un[q] + um[q];
                if (pend) { Pacc[q] += cpend.x * un[q]; Qacc[q] += cpend.y * un[q]; }
            }
        };
        apply(w, mid);
#pragma unroll
        for (int q = 0; q < NS; ++q) um[q] = TRANSPOSED ? e[q] * w[q] - t[q] : w[q] - t[q];
        cpend = cn; pend = true;
    };
    auto flush = [&](const double (&un)[NS]) {
        if (!pend) return;
#pragma unroll
        for (int q = 0; q < NS; ++q) { Pacc[q] += cpend.x * un[q]; Qacc[q] += cpend.y * un[q]; }
    };
    if (order >= 2) step(ua, ub, e1, b, true, c[1]);           // u_2 in ub
    int n = 3;
    for (; n + 1 <= order; n += 2) {
        step(ub, ua, e2, b2, false, c[n - 1]);                  // u_n in ua
        step(ua, ub, e2, b2, false, c[n]);                      // u_{n+1} in ub
    }
    if (n <= order) { step(ub, ua, e2, b2, false, c[n - 1]); flush(ua); }
    else if (order >= 2) flush(ub);
}

template <int P, bool TRANSPOSED, bool UNI, bool ROWS = false>
__device__ __forceinline__ void kpm_series_sq(double (&Pacc)[P * P], double (&Qacc)[P * P], const double (&vin)[P * P],
                                              const double (&eb)[P * P], const double2 *c, int order, double a, double b,
                                              const SqLane<P> &T) {
    kpm_series<P * P, TRANSPOSED>(Pacc, Qacc, vin, eb, c, order, a, b, [&T](double (&w)[P * P], auto &&mid) {
        if constexpr (ROWS) sq16_cb_apply<TRANSPOSED, UNI>(w, T, mid);
        else { mid(); sq_cb_apply<P, TRANSPOSED, UNI>(w, T); }
    });
}

// the GRID layout (cg_fast_common.h: any even-L square lattice up to 16 x 16, uniform hopping): the step's independent work (mid) first,
// then the sweep — four crossing values per crossing colour in one ds_bpermute round trip each
template <bool TRANSPOSED>
__device__ __forceinline__ void kpm_series_grid(double (&Pacc)[4], double (&Qacc)[4], const double (&vin)[4], const double (&eb)[4],
                                                const double2 *c, int order, double a, double b, const GridCtx &T) {
    kpm_series<4, TRANSPOSED>(Pacc, Qacc, vin, eb, c, order, a, b, [&T](double (&w)[4], auto &&mid) {
        mid();
        grid_sweepN<1, TRANSPOSED>(reinterpret_cast<double (&)[1][4]>(w), T);
    });
}

// the HGRID layout (cg_fast_common.h: honeycomb lattices on a grid of lanes, NS = 2, 4 or 8 registers per lane, uniform hopping)
template <int NS, bool TRANSPOSED>
__device__ __forceinline__ void kpm_series_hgrid(double (&Pacc)[NS], double (&Qacc)[NS], const double (&vin)[NS], const double (&eb)[NS],
                                                 const double2 *c, int order, double a, double b, const HgCtx &T) {
    kpm_series<NS, TRANSPOSED>(Pacc, Qacc, vin, eb, c, order, a, b, [&T](double (&w)[NS], auto &&mid) {
        mid();
        hgrid_sweepN<NS, 1, TRANSPOSED>(reinterpret_cast<double (&)[1][NS]>(w), T);
    });
}

template <bool TRANSPOSED>
__device__ __forceinline__ void kpm_series_hc(double (&Pacc)[6], double (&Qacc)[6], const double (&vin)[6], const double (&eb)[6],
                                              const double2 *c, int order, double a, double b, const HcLane &T) {
    kpm_series<6, TRANSPOSED>(Pacc, Qacc, vin, eb, c, order, a, b, [&T](double (&w)[6], auto &&mid) { hc12q_cb_apply<TRANSPOSED>(w, T, mid); });
}

}  // namespace kpmsq
