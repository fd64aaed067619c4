// cg_wg_dev.h — device-side building blocks of the workgroup-resident solvers (cg_wg.hip: k_cg_wg; pcg_wg.hip: k_pcg_wg):
// lane-program and DPP checkerboard sweeps, self-tagged granule stores / polls, wave and workgroup sums, the four-sum records of
// the single-meeting iteration, the mailbox protocol of a sharded solve.  Included by both translation units.
#pragma once
#include "cg_fast_common.h"

namespace wg {

constexpr int MC = 4;                       // colours of the lane program (square / honeycomb / chain lattices)
typedef unsigned long long u64;

template <int NPL>
__host__ __device__ constexpr int slab_len() { return NPL * WAVE + 2 * WAVE; }

struct WgCtl {
    u64 *slots;          // [nrhs][SLOTS_PER_RHS]: four-sum records [32][8 granules] | single-sum records [32][2] (a shard: [p.z | r.r][32][2])
    u64 *bnd;            // [nrhs][G][2][NPL*64][2 granules]: first / last slice of r of every workgroup (G > 1 only)
    int *abort;
    unsigned epoch0;     // tags of this launch are epoch0 + iteration: every launch of a handle gets a range of its own, so the
                         // granules need no zeroing between launches (the host zeroes them when it allocates them and when the
                         // 32-bit range wraps) — a 12 MB fill and its launch boundary less per solve at 288 right-hand sides
    int G, W;
    long long timeout_ticks;   // wall_clock64 ticks (100 MHz)
    long long fixed_iters;     // > 0: measurement mode, exactly this many iterations, no stop test
    int x0_zero;               // the initial guess is zero (the library zeroed it): x0 is not read
    int teams_per_xcd;         // persistent teams: team tq of an XCD takes right-hand sides tq, tq + teams_per_xcd, ... (times 8, plus the XCD)
    const void *ranks = nullptr;   // k_cg_wg<..., RANKS>: WgRankArgs[P] in device memory — the launch arguments of every rank of the grid (nullptr otherwise)
};

template <int NPL>
__device__ __forceinline__ void load_ij(unsigned (&ij)[MC * ((NPL + 1) / 2)], const ModelDev &m, int lane) {
#pragma unroll
    for (int e = 0; e < MC * ((NPL + 1) / 2); ++e) ij[e] = m.lp_ij[e * WAVE + lane];
}

// hopping tables of one tau-slice as a lane keeps them: NE (cosh, sinh) pairs, or ONE pair when every bond of the lattice has
// the same hopping (UNI: no disorder — the example decks; 4*NE registers less).  With UNI the idle lane-program slots (ragged
// colours) transform their private padding pair with the real (c, s) instead of (1, 0): garbage in, garbage out, never read.
template <int NE, bool UNI>
struct Tab {
    double c[UNI ? 1 : NE], s[UNI ? 1 : NE];
    __device__ __forceinline__ double C(int e) const { return c[UNI ? 0 : e]; }
    __device__ __forceinline__ double S(int e) const { return s[UNI ? 0 : e]; }
};

template <int NE, bool UNI>
__device__ __forceinline__ void load_tab(Tab<NE, UNI> &t, const double *lc, const double *ls, int lane, const ModelDev &m) {
    if (UNI) { t.c[0] = m.c_uni; t.s[0] = m.s_uni; return; }
#pragma unroll
    for (int e = 0; e < (UNI ? 1 : NE); ++e) { t.c[e] = lc[e * WAVE + lane]; t.s[e] = ls[e * WAVE + lane]; }
}

// Checkerboard sweep (Checkerboard.jl:57-83 forward / :149-175 reverse) on NS independent slabs at once, bonds in registers:
// the slabs' colour stages interleave, so NS sweeps cost the latency of one.  Slab k lives at buf + k * SL and uses the hopping
// tables tabs[TSTRIDE * k + T0] (SSH: one table set per slice; otherwise TSTRIDE = 0).
template <int NPL, int NS, bool REVERSE, bool UNI, int NT, int TSTRIDE, int T0>
__device__ __forceinline__ void sweepN(double *buf, const unsigned (&ij)[MC * ((NPL + 1) / 2)],
                                       const Tab<MC * ((NPL + 1) / 2), UNI> (&tabs)[NT], int ncol) {
    constexpr int PP = (NPL + 1) / 2, SL = slab_len<NPL>();
#pragma unroll
    for (int cc = 0; cc < MC; ++cc) {
        const int col = REVERSE ? MC - 1 - cc : cc;
        if (col < ncol) {
            double a0[NS][PP], a1[NS][PP];
#pragma unroll
            for (int k = 0; k < NS; ++k)
#pragma unroll
                for (int pp = 0; pp < PP; ++pp) {
                    const unsigned w = ij[col * PP + pp];
                    a0[k][pp] = buf[k * SL + (w & 0xFFFF)]; a1[k][pp] = buf[k * SL + (w >> 16)];
                }
#pragma unroll
            for (int k = 0; k < NS; ++k)
#pragma unroll
                for (int pp = 0; pp < PP; ++pp) {
                    const int e = col * PP + pp;
                    const unsigned w = ij[e];
                    const Tab<MC * ((NPL + 1) / 2), UNI> &t = tabs[TSTRIDE * k + T0];
                    buf[k * SL + (w & 0xFFFF)] = t.C(e) * a0[k][pp] + t.S(e) * a1[k][pp];
                    buf[k * SL + (w >> 16)] = t.C(e) * a1[k][pp] + t.S(e) * a0[k][pp];
                }
            WAVE_LDS_ORDER();
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// The 16 x 16 square lattice with the reference's colouring [x-even | x-odd | y-even | y-odd] (verified on the host:
// detect_square): the checkerboard WITHOUT LDS slabs.  Lane l holds a 2 x 2 PATCH of sites:
//     X = (l >> 1) & 7,  Y = 2 (l >> 4) + (l & 1):  x = 2 X + (q & 1),  y = 2 Y + (q >> 1)   (registers q = 0..3)
// with the two rows of the patch stored in REVERSE order in the lanes of odd Y (q >> 1 = 0 is the row y = 2 Y + 1 there).  Then
//   x-even, y-even  pair two registers of one lane                                   — no data movement at all;
//   x-odd           pairs (q odd) with the lane 2 up in the 16-lane DPP row and (q even) with the lane 2 down (patches X +- 1 of
//                   the same Y sit 2 lanes apart, cyclically — exactly the period of row_ror): 4 DPP moves of an f64;
//   y-odd           registers 2, 3 (the upper row of an even Y, the lower row of an odd Y — the same registers thanks to the
//                   reversed storage) swap with the neighbouring lane of the quad pair: 2 DPP moves of an f64;  registers 0, 1
//                   cross to the next / previous group of 16 lanes: one ds_bpermute pair each.
// A sweep is 12 DPP moves + 4 ds_bpermute + 16 fma per slab (the column-segment layout it replaces: 24 + 4 + 16).
// ------------------------------------------------------------------------------------------------------------------------
// Hopping of the DPP form.  Disordered hopping: the (cosh, sinh) of the bond that covers each of the lane's four sites in each of
// the four colours, gathered once before the loop.  UNI (one hopping for every bond — the example decks): every site has exactly
// one bond per colour, so a colour is  c (I + th P_colour)  with th = sinh/cosh, and a sweep  c^4 prod_colours (I + th P_colour):
// the sweep applies the bracket (ONE fma per site and colour instead of mul + fma) and hands the factor k4 = c^4 to the caller,
// who folds it into the constant of the fma that consumes the swept vector.  Same operator in real arithmetic; against the
// reference's  c y_i + s y_j  it differs by rounding only (a few ulp per sweep; parity tolerances in tests/ unchanged).
// f64 instructions are what the two waves of a SIMD compete for in this kernel (profiles/r02/wg_phase_stamps.log).
// (SSH — one table set per time slice: SqSsh below.  Round 2 built it with a (cosh, sinh) per site and colour, one slice per wave,
// under the two-meeting iteration whose team of 20 workgroups set the time — no gain then.)
template <bool UNI>
struct SqCtx {
    static constexpr int ARITH = UNI ? 1 : 2;                 // vector-ALU instructions per site update
    static constexpr bool GROUPS = true;                      // the y-odd colour is laid out with scheduling groups (see sq_colour)
    double c[UNI ? 1 : 4][UNI ? 1 : 4], s[UNI ? 1 : 4][UNI ? 1 : 4];   // UNI: s[0][0] = th, c[0][0] unused
    double k4;                                                // factor the caller applies to a swept vector (1 unless UNI)
    int yx;                                                   // partner lane of the crossing half of the y-odd colour: (l + 15) & 63 for odd Y, (l + 49) & 63 for even Y
    // new value of a site with value v whose partner holds t
    __device__ __forceinline__ double upd(int col, int k, double v, double t) const {
        if constexpr (UNI) return v + s[0][0] * t;
        else return __builtin_fma(s[UNI ? 0 : col][UNI ? 0 : k], t, c[UNI ? 0 : col][UNI ? 0 : k] * v);      // (spelled out: see SqSsh::up_set)
    }
    // (slab n of the sweep: every slab has the same hopping)
    __device__ __forceinline__ double up(int n, int col, int k, double v, double t) const { return upd(col, k, v, t); }
};

// Bond phonons (SSH) on the same lattice: EVERY time slice has its own hopping, a wave that owns T slices sweeps with T + 1 table
// sets.  A set, as a lane keeps it: ONE (cosh, sinh) per register pair for the two colours that pair registers of the lane, one per
// register for the two crossing colours — 24 doubles (a (cosh, sinh) per site and colour would be 32).  Two sets fit the register
// file next to the Krylov vectors; at 2 slices per wave the set of slice t0 (used by the forward sweep only; the sets of t0+1, t0+2
// serve the forward and the reverse sweep) lives in LDS, [24][64] lane-linear per wave.
struct SqTabS {
    double ci[2][2], si[2][2];     // [x-even | y-even][pair]: x-even pairs registers (0,1), (2,3); y-even pairs (0,2), (1,3)
    double cx[2][4], sx[2][4];     // [x-odd | y-odd][register]
};
constexpr int SQ_TABS = 24;        // doubles per lane and set; LDS order: ci[0], si[0], ci[1], si[1], cx[0], sx[0], cx[1], sx[1]

template <int NREG, bool LDS0>
struct SqSsh {
    SqTabS t[NREG];                // LDS0: t[k] = set k + 1 (set 0 in LDS); otherwise t[k] = set k
    const double *l0;              // LDS0: this lane's column of set 0 (entry e at l0[e * 64])
    int yx;
    __device__ __forceinline__ double up_set(int set, int col, int k, double v, double tv) const {
        double c, s;
        if (LDS0 && set == 0) {
            const int e = (col == 0) ? (k >> 1) : (col == 2) ? 4 + (k & 1) : (col == 1) ? 8 + k : 16 + k;
            c = l0[e * WAVE]; s = l0[(e + ((col & 1) ? 4 : 2)) * WAVE];
        } else {
            const SqTabS &T = t[LDS0 ? set - 1 : set];
            if (col == 0)      { c = T.ci[0][k >> 1]; s = T.si[0][k >> 1]; }
            else if (col == 2) { c = T.ci[1][k & 1];  s = T.si[1][k & 1]; }
            else               { c = T.cx[col >> 1][k]; s = T.sx[col >> 1][k]; }
        }
        // (the contraction is spelled out — round(c v), then one fma: left to -ffp-contract the compiler picks which product it fuses, and
        //  two builds of this file, e.g. the ELPH_LDS_SYNC A/B library, need not pick the same)
        return __builtin_fma(s, tv, c * v);
    }
};
// slab n of a sweep uses table set SOFF + n (forward sweep of w(t0 .. t0+T): 0; reverse sweep of w(t0+1 ..): 1)
template <int NREG, bool LDS0, int SOFF>
struct SqSshView {
    static constexpr int ARITH = 2;
    static constexpr bool GROUPS = false;
    const SqSsh<NREG, LDS0> &X;
    int yx;
    __device__ __forceinline__ double up(int n, int col, int k, double v, double t) const { return X.up_set(SOFF + n, col, k, v, t); }
};

// One colour on CNT (1 or 2) slabs (slabs N0, N0 + 1 of the sweep), as one scheduling region: the cross-lane moves of the slabs
// first, then their arithmetic.  Left to itself the scheduler (256 registers, none to spare) funnels every ds_bpermute through ONE
// temporary and waits for each; with every slab's moves hoisted to the front it spills instead.
template <int CNT, int COL, int N0, class CTX>
__device__ __forceinline__ void sq_colour(double (*v)[4], const CTX &X) {
    constexpr int DS = 0x080, VALU = 0x002, ARITH = CTX::ARITH;
    if constexpr (COL == 0) {                                // x even: (0,1), (2,3) of the lane itself
#pragma unroll
        for (int n = 0; n < CNT; ++n) {
            const double n0 = X.up(N0 + n, 0, 0, v[n][0], v[n][1]), n1 = X.up(N0 + n, 0, 1, v[n][1], v[n][0]);
            const double n2 = X.up(N0 + n, 0, 2, v[n][2], v[n][3]), n3 = X.up(N0 + n, 0, 3, v[n][3], v[n][2]);
            v[n][0] = n0; v[n][1] = n1; v[n][2] = n2; v[n][3] = n3;
        }
    } else if constexpr (COL == 1) {                         // x odd: q odd <-> q - 1 of the lane 2 up, q even <-> q + 1 of the lane 2 down
#pragma unroll
        for (int n = 0; n < CNT; ++n) {
            const double t1 = dpp_f64<0x12E>(v[n][0]), t3 = dpp_f64<0x12E>(v[n][2]);     // row_ror:14 = lane + 2
            const double t0 = dpp_f64<0x122>(v[n][1]), t2 = dpp_f64<0x122>(v[n][3]);     // row_ror:2  = lane - 2
            v[n][0] = X.up(N0 + n, 1, 0, v[n][0], t0); v[n][1] = X.up(N0 + n, 1, 1, v[n][1], t1);
            v[n][2] = X.up(N0 + n, 1, 2, v[n][2], t2); v[n][3] = X.up(N0 + n, 1, 3, v[n][3], t3);
        }
    } else if constexpr (COL == 2) {                         // y even: (0,2), (1,3) of the lane itself
#pragma unroll
        for (int n = 0; n < CNT; ++n) {
            const double n0 = X.up(N0 + n, 2, 0, v[n][0], v[n][2]), n2 = X.up(N0 + n, 2, 2, v[n][2], v[n][0]);
            const double n1 = X.up(N0 + n, 2, 1, v[n][1], v[n][3]), n3 = X.up(N0 + n, 2, 3, v[n][3], v[n][1]);
            v[n][0] = n0; v[n][1] = n1; v[n][2] = n2; v[n][3] = n3;
        }
    } else {                                                 // y odd: 2, 3 swap inside the lane pair; 0, 1 cross to the neighbouring row group
        double c0[CNT], c1[CNT];
#pragma unroll
        for (int n = 0; n < CNT; ++n) { c0[n] = __shfl(v[n][0], X.yx, WAVE); c1[n] = __shfl(v[n][1], X.yx, WAVE); }
#pragma unroll
        for (int n = 0; n < CNT; ++n) {
            const double t2 = dpp_f64<0xB1>(v[n][2]), t3 = dpp_f64<0xB1>(v[n][3]);       // quad_perm [1,0,3,2]
            v[n][2] = X.up(N0 + n, 3, 2, v[n][2], t2); v[n][3] = X.up(N0 + n, 3, 3, v[n][3], t3);
        }
#pragma unroll
        for (int n = 0; n < CNT; ++n) { v[n][0] = X.up(N0 + n, 3, 0, v[n][0], c0[n]); v[n][1] = X.up(N0 + n, 3, 1, v[n][1], c1[n]); }
        if constexpr (CTX::GROUPS) {
            __builtin_amdgcn_sched_group_barrier(DS, 4 * CNT, 0);
            __builtin_amdgcn_sched_group_barrier(VALU, (4 + 4 * ARITH) * CNT, 0);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
}

template <int NS, int N0, int COL, class CTX>
__device__ __forceinline__ void sq_pairs(double (&v)[NS][4], const CTX &X) {
    sq_colour<(N0 + 1 < NS) ? 2 : 1, COL, N0, CTX>(&v[N0], X);
    if constexpr (N0 + 2 < NS) sq_pairs<NS, N0 + 2, COL, CTX>(v, X);
}

template <int NS, bool REVERSE, class CTX>
__device__ __forceinline__ void sq_sweepC(double (&v)[NS][4], const CTX &X) {
    if constexpr (!REVERSE) { sq_pairs<NS, 0, 0, CTX>(v, X); sq_pairs<NS, 0, 1, CTX>(v, X); sq_pairs<NS, 0, 2, CTX>(v, X); sq_pairs<NS, 0, 3, CTX>(v, X); }
    else                    { sq_pairs<NS, 0, 3, CTX>(v, X); sq_pairs<NS, 0, 2, CTX>(v, X); sq_pairs<NS, 0, 1, CTX>(v, X); sq_pairs<NS, 0, 0, CTX>(v, X); }
}

template <int NS, bool REVERSE, bool UNI>
__device__ __forceinline__ void sq_sweepN(double (&v)[NS][4], const SqCtx<UNI> &X) { sq_sweepC<NS, REVERSE, SqCtx<UNI>>(v, X); }

// SSH: slab n with table set SOFF + n
template <int NS, bool REVERSE, int SOFF, int NREG, bool LDS0>
__device__ __forceinline__ void sq_sweepS(double (&v)[NS][4], const SqSsh<NREG, LDS0> &X) {
    const SqSshView<NREG, LDS0, SOFF> V{X, X.yx};
    sq_sweepC<NS, REVERSE, SqSshView<NREG, LDS0, SOFF>>(v, V);
}

// ------------------------------------------------------------------------------------------------------------------------
// The honeycomb lattice of 12 x 12 cells (config D: 288 sites) in the reference's colouring [A-B of a cell | B(x,y)-A(x+1,y) |
// B(x,y)-A(x,y+1)] (verified on the host: detect_honeycomb12): the checkerboard WITHOUT LDS slabs.  A 16-lane DPP row holds the
// twelve cells x = 0..11 of three lattice rows in its lanes 2..13, one lane per x, plus TWO MIRROR LANES on either side — lanes 0, 1
// carry copies of x = 10, 11 and lanes 14, 15 copies of x = 0, 1 — so that the x-neighbour is always the next lane of the row
// (row_ror:1 / row_ror:15) although 12 is no period of any DPP pattern.  Row g of the wave (lanes 16 g ..) holds the lattice rows
// y = 3 g + j, a lane the six sites q = 2 j + orbital of its three cells:
//   A-B of a cell        pairs registers (0,1), (2,3), (4,5) of the lane                    — no data movement;
//   B(x,y)-A(x+1,y)      the A registers take B of the lane below, the B registers A of the lane above: 6 DPP moves of an f64;
//   B(x,y)-A(x,y+1)      pairs (1,2), (3,4) of the lane; register 5 crosses to register 0 of the next row of the wave: one
//                        ds_bpermute pair each way.
// The mirror lanes do everything the real lanes do, on copies.  A colour-2 step spoils the outermost correct lane on either side
// (lane 0 has no x = 9 below it); a mat-vec holds two such steps (forward sweep, reverse sweep), so lanes 2..13 — the real ones —
// end it exact, the mirrors not.  The mirrors are never repaired: every vector update is pointwise, p of a mirror lane is made
// from the REAL lane's residual (read from LDS at the real lane's slot), so p, the only input of the next mat-vec, is an exact
// copy again.  Mirror lanes stay out of the inner products and of the stores to memory.  25 % of the lanes work twice — the vector
// ALU is not what bounds this kernel — against 0.4 us of LDS traffic per sweep in the lane-program form.
// Uniform hopping only (one (cosh, sinh) for all bonds: the example decks): a colour is c (I + th P), a sweep c^3 prod (I + th P).
// (The QUAD layout of the Chebyshev recursion — kpm_sq_dev.h: three cells of a row per lane, four lanes of a DPP quad per row, no
// mirrors, 4 DPP moves + 12 ds_bpermute per slab on 48 lanes — was built into this kernel too and is SLOWER here: 65.1 against 60.2 us
// per iteration of 256 right-hand sides, 4.27 against 3.82 at 8 (profiles/r03/time_forms_D_quad_layout_rejected.log): eight waves of a
// CU share ONE LDS crossbar, a recursion's single wave per SIMD does not.)
// ------------------------------------------------------------------------------------------------------------------------
constexpr int HC_NPL = 6;
__device__ __forceinline__ int hc_site(int lane, int q) {
    const int c = lane & 15, x = (c >= 2) ? ((c >= 14) ? c - 14 : c - 2) : c + 10;
    return 2 * (x + 12 * (3 * (lane >> 4) + (q >> 1))) + (q & 1);
}
__device__ __forceinline__ bool hc_real(int lane) { const int c = lane & 15; return c >= 2 && c <= 13; }
// the real lane whose copy a mirror lane carries (a real lane: itself)
__device__ __forceinline__ int hc_src_lane(int lane) { const int c = lane & 15; return (c < 2) ? lane + 12 : ((c > 13) ? lane - 12 : lane); }

struct HcCtx {
    double th, k3;             // tanh of the bond angle; the factor c^3 the caller applies to a swept vector
    int up, dn;                // lane + 16, lane - 16 (cyclic in the wave): the next / previous row
};

template <int CNT, int COL>
__device__ __forceinline__ void hc_colour(double (*v)[HC_NPL], const HcCtx &X) {
    if constexpr (COL == 0) {
#pragma unroll
        for (int n = 0; n < CNT; ++n)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const double a = v[n][2 * j] + X.th * v[n][2 * j + 1], b = v[n][2 * j + 1] + X.th * v[n][2 * j];
                v[n][2 * j] = a; v[n][2 * j + 1] = b;
            }
    } else if constexpr (COL == 1) {
#pragma unroll
        for (int n = 0; n < CNT; ++n) {
            double ta[3], tb[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) { ta[j] = dpp_f64<0x121>(v[n][2 * j + 1]); tb[j] = dpp_f64<0x12F>(v[n][2 * j]); }   // row_ror:1 = lane - 1, row_ror:15 = lane + 1
#pragma unroll
            for (int j = 0; j < 3; ++j) { v[n][2 * j] += X.th * ta[j]; v[n][2 * j + 1] += X.th * tb[j]; }
        }
    } else {
        double c0[CNT], c5[CNT];
#pragma unroll
        for (int n = 0; n < CNT; ++n) { c0[n] = __shfl(v[n][5], X.dn, WAVE); c5[n] = __shfl(v[n][0], X.up, WAVE); }
#pragma unroll
        for (int n = 0; n < CNT; ++n)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const double b = v[n][2 * j + 1] + X.th * v[n][2 * j + 2], a = v[n][2 * j + 2] + X.th * v[n][2 * j + 1];
                v[n][2 * j + 1] = b; v[n][2 * j + 2] = a;
            }
#pragma unroll
        for (int n = 0; n < CNT; ++n) { v[n][0] += X.th * c0[n]; v[n][5] += X.th * c5[n]; }
    }
    __builtin_amdgcn_sched_barrier(0);
}

template <int NS, int N0, int COL>
__device__ __forceinline__ void hc_pairs(double (&v)[NS][HC_NPL], const HcCtx &X) {
    hc_colour<(N0 + 1 < NS) ? 2 : 1, COL>(&v[N0], X);
    if constexpr (N0 + 2 < NS) hc_pairs<NS, N0 + 2, COL>(v, X);
}

template <int NS, bool REVERSE>
__device__ __forceinline__ void hc_sweepN(double (&v)[NS][HC_NPL], const HcCtx &X) {
    if constexpr (!REVERSE) { hc_pairs<NS, 0, 0>(v, X); hc_pairs<NS, 0, 1>(v, X); hc_pairs<NS, 0, 2>(v, X); }
    else                    { hc_pairs<NS, 0, 2>(v, X); hc_pairs<NS, 0, 1>(v, X); hc_pairs<NS, 0, 0>(v, X); }
}

// ------------------------------------------------------------------------------------------------------------------------
// The 8 x 8 square lattice (config B: 64 sites, ONE per lane) in the reference's colouring [x-even | x-odd | y-even | y-odd] (verified
// on the host: detect_square, sq_P = 1): the checkerboard without LDS slabs.  Lane l holds the site x = (l >> 1) & 7,
// y = 2 (l >> 4) + (l & 1): two lattice rows are interleaved in a 16-lane DPP row, so that x +- 1 is the lane 2 up / 2 down of the
// row, cyclically (the period of row_ror, as on the 16 x 16 lattice):
//   x-even  partner = lane ^ 2: quad_perm [2,3,0,1];            x-odd   the lane 2 up (x odd) or 2 down (x even): two row
//   y-even  partner = lane ^ 1: quad_perm [1,0,3,2];                    rotations and a select;
//   y-odd   the next / previous 16-lane row, other parity (sq_patch_ycross): one ds_bpermute pair.
// 10 moves + 1 ds_bpermute pair + 4 fma per slab.  Uniform hopping: a colour is c (I + th P), the caller applies c^4.
// ------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int s8_site(int lane) { return ((lane >> 1) & 7) + 8 * (2 * (lane >> 4) + (lane & 1)); }
struct S8Ctx { double th, k4; int yx; bool xodd; };

template <int NS, bool REVERSE>
__device__ __forceinline__ void s8_sweepN(double (&v)[NS][1], const S8Ctx &X) {
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
        const int col = REVERSE ? 3 - cc : cc;
        if (col == 0) {
#pragma unroll
            for (int n = 0; n < NS; ++n) v[n][0] += X.th * dpp_f64<0x4E>(v[n][0]);
        } else if (col == 1) {
#pragma unroll
            for (int n = 0; n < NS; ++n) {
                const double up = dpp_f64<0x12E>(v[n][0]), dn = dpp_f64<0x122>(v[n][0]);      // row_ror:14 = lane + 2, row_ror:2 = lane - 2
                v[n][0] += X.th * (X.xodd ? up : dn);
            }
        } else if (col == 2) {
#pragma unroll
            for (int n = 0; n < NS; ++n) v[n][0] += X.th * dpp_f64<0xB1>(v[n][0]);
        } else {
            double t[NS];
#pragma unroll
            for (int n = 0; n < NS; ++n) t[n] = __shfl(v[n][0], X.yx, WAVE);
#pragma unroll
            for (int n = 0; n < NS; ++n) v[n][0] += X.th * t[n];
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// Diagnostic build (-DELPH_WG_STAMPS, tools/time_wg_phases.py): wave 0 of workgroup 0 of right-hand side 0 adds the wall-clock
// ticks (100 MHz) it spends in each phase of an iteration to a buffer no kernel reads.  Never compiled into the product.
#ifdef ELPH_WG_STAMPS
__device__ unsigned long long g_wg_stamps[16];
#define STAMP_DECL long long _ts = wall_clock64(); unsigned long long _acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define STAMP(k) do { __builtin_amdgcn_sched_barrier(0); const long long _n = wall_clock64(); _acc[k] += (unsigned long long)(_n - _ts); _ts = _n; __builtin_amdgcn_sched_barrier(0); } while (0)
#ifndef ELPH_WG_STAMP_WAVE
#define ELPH_WG_STAMP_WAVE 0
#endif
#define STAMP_OUT(iters) do { if (rhs == 0 && g == 0 && wv == ELPH_WG_STAMP_WAVE && lane == 0) { for (int _k = 0; _k < 10; ++_k) g_wg_stamps[_k] = _acc[_k]; g_wg_stamps[10] = (unsigned long long)(iters); } } while (0)
#else
#define STAMP_DECL
#define STAMP(k)
#define STAMP_OUT(iters)
#endif

// One 8-byte record granule, write-through (sc1): correct under any placement of the team's workgroups.  (Measured alternative
// for teams that find themselves on one XCD — plain stores that stay in that XCD's L2, polled with the same sc1 loads — is
// SLOWER, 9.3 vs 6.7 us per iteration at config C: a plain store is in no hurry to leave the CU.)
__device__ __forceinline__ void st_gran(u64 *p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u64 ld_gran(const u64 *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_sc1(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld_sc1(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Poll loops of the meetings: lanes < 2G watch the team's record granules, every lane of a boundary wave watches its 2 NPL boundary
// granules, until all carry `epoch`.  A poll is a ~0.5 us round trip to the memory side (write-through lines do not stay in L2),
// so the record poll keeps THREE loads in flight, a fresh one issued as the oldest returns: a record is noticed ~0.15 us after
// it lands instead of up to a round trip later — and the workgroups of a team stay that much closer in step.
// Bounded by the wall clock; false = gave up (abort raised) or saw abort.
template <int NPL>
__device__ __forceinline__ bool poll_bail(int spin, long long &t_start, int lane, const WgCtl &R) {
    if ((spin & 31) != 31) return false;
    if (__hip_atomic_load(R.abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return true;
    const long long now = wall_clock64();
    if (t_start == 0) { t_start = now; return false; }
    if (now - t_start > R.timeout_ticks) {
        if (lane == 0) __hip_atomic_store(R.abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return true;
    }
    return false;
}

// records only (lanes < 2G): three polls in flight
__device__ __forceinline__ bool poll_records(const u64 *rec, int G, unsigned epoch, int lane, const WgCtl &R, u64 &v) {
    const bool mine = lane < 2 * G;
    u64 a = 0, b = 0, c = 0;
    if (mine) a = ld_gran(rec + lane);
    __builtin_amdgcn_s_sleep(2);
    if (mine) b = ld_gran(rec + lane);
    __builtin_amdgcn_s_sleep(2);
    if (mine) c = ld_gran(rec + lane);
    long long t_start = 0;
    for (int spin = 0;; ++spin) {
        if (__all(!mine || (unsigned)(a >> 32) == epoch)) { v = a; return true; }
        if (mine) a = ld_gran(rec + lane);
        if (__all(!mine || (unsigned)(b >> 32) == epoch)) { v = b; return true; }
        if (mine) b = ld_gran(rec + lane);
        if (__all(!mine || (unsigned)(c >> 32) == epoch)) { v = c; return true; }
        if (mine) c = ld_gran(rec + lane);
        if (poll_bail<1>(spin, t_start, lane, R)) return false;
    }
}

// records (optional) + the 2 NPL boundary granules of every lane; one poll at a time (a second poll set in flight costs 4 NPL + 2
// registers next to the Krylov vectors: measured as spills)
template <int NPL>
__device__ __forceinline__ bool poll_granules(const u64 *rec, int G, const u64 *bh, unsigned epoch, int lane, const WgCtl &R, u64 &v,
                                              u64 (&gh)[NPL][2]) {
    long long t_start = 0;
    for (int spin = 0;; ++spin) {
        bool ok = true;
        if (rec && lane < 2 * G) v = ld_gran(rec + lane);
#pragma unroll
        for (int q = 0; q < NPL; ++q) { gh[q][0] = ld_gran(bh + 2 * (lane + q * WAVE)); gh[q][1] = ld_gran(bh + 2 * (lane + q * WAVE) + 1); }
        if (rec && lane < 2 * G) ok = ((unsigned)(v >> 32) == epoch);
#pragma unroll
        for (int q = 0; q < NPL; ++q) ok = ok && (unsigned)(gh[q][0] >> 32) == epoch && (unsigned)(gh[q][1] >> 32) == epoch;
        if (__all(ok)) return true;
        if (poll_bail<NPL>(spin, t_start, lane, R)) return false;
        __builtin_amdgcn_s_sleep(1);
    }
}

// ---- the ONE meeting of an iteration (single-meeting form): a workgroup's record is FOUR sums — p.z, r.z, z.z and r.r — as eight
// self-tagged granules; a team's records fill 8 G granules (G <= 32: up to four loads per lane of the polling wave).
constexpr int REC4 = 8;                      // granules per record
constexpr int SLOTS_A = REC4 * 32;           // granules of a right-hand side's four-sum records (G <= 32)
constexpr int SLOTS_PER_RHS = SLOTS_A + 64;  // + the single-sum records of the fallback meeting (direct r.r)

// the workgroup's record: `mine` = sum_part4 (lanes 8 j .. 8 j + 7 hold value j); lanes 8 j and 8 j + 1 store its two halves
__device__ __forceinline__ void publish_rec4(u64 *slots, int g, double mine, unsigned epoch, int lane) {
    if (lane < 32 && (lane & 7) < 2) {
        const u64 bits = (u64)__double_as_longlong(mine);
        st_gran(slots + REC4 * g + 2 * (lane >> 3) + (lane & 1), ((u64)epoch << 32) | ((lane & 1) ? (bits >> 32) : (bits & 0xFFFFFFFFull)));
    }
}

// The four totals of the team from the polled granules, lane-parallel (no scalar round trips): lane l of load k holds granule
// l of records 8k .. 8k+7 = {record 8k + (l >> 3), value (l & 7) >> 1, half l & 1}.  Halves -> f64 in the even lanes, records of
// the four loads added in load order, then a butterfly over the eight records of a load (lanes 8 apart): every lane with
// (l & 7) == 2 j ends with the total of value j.  One fixed tree, the same in every wave that runs it on the same records.
template <int NL>
__device__ __forceinline__ double sum_rec4(const u64 (&v)[NL], int G, int lane) {
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < NL; ++k) {
        const int lo = (int)(unsigned)v[k];
        const int hi = __builtin_amdgcn_update_dpp(0, lo, 0xF5, 0xF, 0xF, true);      // quad_perm [1,1,3,3]: the odd neighbour's half
        const double val = __hiloint2double(hi, lo);
        acc += (8 * k + (lane >> 3) < G) ? val : 0.0;
    }
    acc += dpp_f64<0x128>(acc);                         // row_ror:8 — records 1 apart (8 lanes)
    acc += __shfl_xor(acc, 16, WAVE);
    acc += __shfl_xor(acc, 32, WAVE);
    return acc;
}

// sum over the W wave partials of value j for j = 0..3 at once: lane 8 j + w reads part[j][w]; a butterfly inside each group of eight
// lanes leaves the total of value j in all eight of them (same tree in every wave)
__device__ __forceinline__ double sum_part4(const double *part, int W, int lane) {
    double v = (lane < 32 && (lane & 7) < W) ? part[lane] : 0.0;
    v += dpp_f64<0xB1>(v);          // quad_perm [1,0,3,2]
    v += dpp_f64<0x4E>(v);          // quad_perm [2,3,0,1]
    v += dpp_f64<0x141>(v);         // row_half_mirror: lanes 4 apart inside a group of eight
    return v;
}

// poll the team's four-sum records (rec != nullptr: NL loads per lane cover 8 G granules) and up to two boundary SEGMENTS (64 values of
// a neighbouring workgroup's boundary slice of z, one per lane: b0 / b1 point at this lane's granule pair, nullptr = none).  The
// boundary slices are 2 NPL segments; the waves of a workgroup share them out (see the meeting), so a wave holds 4-8 registers of
// granules while z and p are live instead of 4 NPL + 4.  One poll at a time.
template <int NL>
__device__ __forceinline__ bool poll_rec4(const u64 *rec, int G, const u64 *b0, const u64 *b1, unsigned epoch, int lane, const WgCtl &R,
                                          u64 (&v)[NL], double &z0, double &z1) {
    const int ngran = REC4 * G;
    u64 a0 = 0, a1 = 0, c0 = 0, c1 = 0;
    long long t_start = 0;
    for (int spin = 0;; ++spin) {
        bool ok = true;
        if (rec) {
#pragma unroll
            for (int k = 0; k < NL; ++k) if (lane + WAVE * k < ngran) v[k] = ld_gran(rec + lane + WAVE * k);
        }
        if (b0) { a0 = ld_gran(b0); a1 = ld_gran(b0 + 1); }
        if (b1) { c0 = ld_gran(b1); c1 = ld_gran(b1 + 1); }
        if (rec) {
#pragma unroll
            for (int k = 0; k < NL; ++k) if (lane + WAVE * k < ngran) ok = ok && (unsigned)(v[k] >> 32) == epoch;
        }
        if (b0) ok = ok && (unsigned)(a0 >> 32) == epoch && (unsigned)(a1 >> 32) == epoch;
        if (b1) ok = ok && (unsigned)(c0 >> 32) == epoch && (unsigned)(c1 >> 32) == epoch;
        if (__all(ok)) break;
        if (poll_bail<1>(spin, t_start, lane, R)) return false;
        __builtin_amdgcn_s_sleep(1);
    }
    z0 = __hiloint2double((int)(unsigned)a1, (int)(unsigned)a0);
    z1 = __hiloint2double((int)(unsigned)c1, (int)(unsigned)c0);
    return true;
}

__device__ __forceinline__ void st_f64_gran(u64 *g2, double v, unsigned epoch) {
    const u64 bits = (u64)__double_as_longlong(v), tag = (u64)epoch << 32;
    st_gran(g2, tag | (bits & 0xFFFFFFFFull));
    st_gran(g2 + 1, tag | (bits >> 32));
}

// wave-wide sum without LDS: quad swaps and row mirrors (DPP) give every lane the sum of its 16-lane row, then the four row sums
// are added in a fixed order through scalar registers.  Same tree in every wave and every run.
__device__ __forceinline__ double wave_sum_dpp(double v) {
    v += dpp_f64<0xB1>(v);          // quad_perm [1,0,3,2]
    v += dpp_f64<0x4E>(v);          // quad_perm [2,3,0,1]
    v += dpp_f64<0x141>(v);         // row_half_mirror
    v += dpp_f64<0x140>(v);         // row_mirror
    const double r0 = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 0), __builtin_amdgcn_readlane(__double2loint(v), 0));
    const double r1 = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 16), __builtin_amdgcn_readlane(__double2loint(v), 16));
    const double r2 = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 32), __builtin_amdgcn_readlane(__double2loint(v), 32));
    const double r3 = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 48), __builtin_amdgcn_readlane(__double2loint(v), 48));
    return (r0 + r1) + (r2 + r3);
}

// FOUR wave-wide sums at once, transposed on the way: after two exchange steps inside each quad a lane carries ONE of the four
// values (index lane & 3), then rotations by 4 and 8 inside the 16-lane row and two wave shuffles add the lanes of that index.
// ~35 vector-ALU instructions and 4 ds_bpermute against 4 x 23 for four separate sums.  The result is taken from lanes 0..3
// (value lane & 3): one fixed tree, the same in every wave and every run.
__device__ __forceinline__ double wave_sum4(double a0, double a1, double a2, double a3, int lane) {
    const bool odd = (lane & 1) != 0, hi2 = (lane & 2) != 0;
    double k0 = odd ? a1 : a0, k1 = odd ? a3 : a2;            // kept; the other two go to lane ^ 1
    const double s0 = odd ? a0 : a1, s1 = odd ? a2 : a3;
    k0 += dpp_f64<0xB1>(s0);                                  // quad_perm [1,0,3,2]
    k1 += dpp_f64<0xB1>(s1);
    double k = hi2 ? k1 : k0;
    const double s = hi2 ? k0 : k1;
    k += dpp_f64<0x4E>(s);                                    // quad_perm [2,3,0,1]: lane & 3 = value index from here on
    k += dpp_f64<0x124>(k);                                   // row_ror:4
    k += dpp_f64<0x128>(k);                                   // row_ror:8
    k += __shfl_xor(k, 16, WAVE);
    k += __shfl_xor(k, 32, WAVE);
    return k;
}

__device__ __forceinline__ double readlane_f64(double v, int l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

// sum of the W wave partials of a workgroup in index order (every wave gets the same bits)
__device__ __forceinline__ double wg_sum(const double *part, int W, int lane) {
    const double mine = (lane < W) ? part[lane] : 0.0;
    double tot = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) if (i < W) tot += readlane_f64(mine, i);
    return tot;
}

// wait for this wave's LDS traffic, then the workgroup barrier — NOT __syncthreads(): that would also drain the vector-memory
// queue (the x loads / stores in flight ride across the meetings on purpose)
__device__ __forceinline__ void wg_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// ------------------------------------------------------------------------------------------------------------------------
// One solve over several GPUs (slabs of rows of cells along l2, one process per GPU).  The rank's handle lives on
// its slab = own rows + the ghost rows the fused M^T M needs (the dependency closure of the checkerboard, computed by the
// caller); the SAME resident kernel runs on it, with three differences:
//   * the two inner products count own sites only, and a team's records come from the workgroups of ALL ranks: every
//     workgroup stores its record into every rank's MAILBOX (device memory of that rank, mapped here through hipIpc — a
//     device-initiated store over xGMI, no collective, no host) and polls its own mailbox; all ranks add the same P G
//     records in the same order, so alpha, beta and the stop decision are bit-identical everywhere;
//   * after the residual update every wave stores the values of its slice on the rows its neighbours hold as ghosts into
//     the neighbour's mailbox (granules: the data is its own flag) and takes its own ghost rows of r from its mailbox
//     together with the second meeting — ONE exchange of the checkerboard boundary rows per iteration;
//   * the solve starts from x = 0 inside the kernel (r0 = p0 = b, |b|^2 by a first meeting): no host-side combination.
// Mailbox (identical layout on every rank): [2 meetings][ELPH_SHARD_MAXREC = 256 records][2 granules] then ghost rows from below / from above
// [Ltau][cap_ghost][2 granules] each, the pair once per parity of the iteration (k_cg_wg adds the parity's offset itself).  It is zeroed by elph_shard_prepare; the caller's barrier between prepare and solve
// keeps a fast rank's first stores from being wiped.
// ------------------------------------------------------------------------------------------------------------------------
using ShardCtl = ElphShardCtl;                   // elph_internal.h
constexpr int SH_MAXREC = ELPH_SHARD_MAXREC;                              // P * G records at most
constexpr size_t SH_REC_WORDS = 2 * (size_t)SH_MAXREC * 2;

__device__ __forceinline__ void st_sys(u64 *p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ u64 ld_sys(const u64 *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
// mailbox accesses of a sharded solve.  LOCAL: all ranks are slabs on THIS device (slabs.hip, k_cg_wg<..., RANKS>) and their mailboxes ordinary
// device memory — agent scope, as the records of an un-sharded team (st_gran / ld_gran); otherwise another GPU may be the writer or the
// reader: system scope on the uncached fine-grained mailbox
template <bool LOCAL> __device__ __forceinline__ void st_mail(u64 *p, u64 v) { if constexpr (LOCAL) st_gran(p, v); else st_sys(p, v); }
template <bool LOCAL> __device__ __forceinline__ u64 ld_mail(const u64 *p) { if constexpr (LOCAL) return ld_gran(p); else return ld_sys(p); }
__device__ __forceinline__ u64 *sh_ghost(u64 *mail, int region, int L, int cap, int t, int k) {
    return mail + SH_REC_WORDS + (((size_t)region * L + t) * cap + k) * 2;
}

// wave 0 of a workgroup: store this workgroup's record of meeting m into every rank's mailbox
template <bool LOCAL = false>
__device__ __forceinline__ void sh_publish(const ShardCtl &Sh, int m, int g, int G, double mine, unsigned epoch, int lane) {
    if (lane < 2 * Sh.P) {
        const u64 bits = (u64)__double_as_longlong(mine);
        const int dest = lane >> 1, half = lane & 1;
        st_mail<LOCAL>(Sh.mail[dest] + ((size_t)m * SH_MAXREC + (size_t)Sh.rank * G + g) * 2 + half,
               ((u64)epoch << 32) | (half ? (bits >> 32) : (bits & 0xFFFFFFFFull)));
    }
}

// poll the P G records of meeting m in the own mailbox (wave 0: rec = true) and this lane's ghost granules (gaddr[q] != nullptr);
// on success `total` = sum of the records in (rank, workgroup) order and gv[q] = the ghost values.
// Up to SH_MAXREC = 256 records = 512 granules: eight per lane (8 ranks x 20 workgroups at Ltau = 160 are 320 of them).
template <int NPL, bool LOCAL = false>
__device__ __forceinline__ bool sh_poll(const ShardCtl &Sh, bool rec, int m, int G, const u64 *const (&gaddr)[NPL], unsigned epoch,
                                        int lane, const WgCtl &R, double &total, double (&gv)[NPL]) {
    constexpr int NV = 2 * SH_MAXREC / WAVE;
    const int nrec2 = 2 * Sh.P * G;
    const u64 *rbase = Sh.mail[Sh.rank] + (size_t)m * SH_MAXREC * 2;
    u64 v[NV], g0[NPL], g1[NPL];
#pragma unroll
    for (int s = 0; s < NV; ++s) v[s] = 0;
    long long t_start = 0;
    for (int spin = 0;; ++spin) {
        bool ok = true;
        if (rec) {
#pragma unroll
            for (int s = 0; s < NV; ++s) if (lane + WAVE * s < nrec2) v[s] = ld_mail<LOCAL>(rbase + lane + WAVE * s);
        }
#pragma unroll
        for (int q = 0; q < NPL; ++q) if (gaddr[q]) { g0[q] = ld_mail<LOCAL>(gaddr[q]); g1[q] = ld_mail<LOCAL>(gaddr[q] + 1); }
        if (rec) {
#pragma unroll
            for (int s = 0; s < NV; ++s) if (lane + WAVE * s < nrec2) ok = ok && (unsigned)(v[s] >> 32) == epoch;
        }
#pragma unroll
        for (int q = 0; q < NPL; ++q) if (gaddr[q]) ok = ok && (unsigned)(g0[q] >> 32) == epoch && (unsigned)(g1[q] >> 32) == epoch;
        if (__all(ok)) break;
        if (poll_bail<NPL>(spin, t_start, lane, R)) return false;
        __builtin_amdgcn_s_sleep(1);
    }
    total = 0.0;
    if (rec) {
        const int nr = Sh.P * G;
#pragma unroll
        for (int s = 0; s < NV; ++s) {
            if (s * (WAVE / 2) >= nr) break;
            const int half = (int)(unsigned)v[s];
            for (int k = 0; k < WAVE / 2; ++k)
                if (s * (WAVE / 2) + k < nr) total += __hiloint2double(__builtin_amdgcn_readlane(half, 2 * k + 1), __builtin_amdgcn_readlane(half, 2 * k));
        }
    }
#pragma unroll
    for (int q = 0; q < NPL; ++q) gv[q] = gaddr[q] ? __hiloint2double((int)(unsigned)g1[q], (int)(unsigned)g0[q]) : 0.0;
    return true;
}


}  // namespace wg
