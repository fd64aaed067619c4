// cg_fast_common.h — helpers shared by the two compilations of cg_fast_impl.inc (cg_fast.hip: 4-colour lane programs,
// cg_fast6.hip: 6-colour ones; separate translation units so that they build in parallel).
#pragma once
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

// ---- DPP moves of f64 values (cg_wg.hip, the 16 x 16 Chebyshev kernel) -------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    // (bound_ctrl set: every control used here reads a valid lane, and the destination then needs no initialising move)
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

// The 2 x 2 patch layout of the 16 x 16 square lattice (cg_wg.hip: the DPP checkerboard; cg_fast_impl.inc: the Chebyshev recursion):
// lane l holds the patch X = (l >> 1) & 7, Y = 2 (l >> 4) + (l & 1); register q is the site x = 2 X + (q & 1), y = 2 Y + (q >> 1),
// with the two rows of the patch stored in reverse order in the lanes of odd Y.  Returns x + 16 y.
__host__ __device__ __forceinline__ int sq_patch_site(int lane, int q) {
    const int X = (lane >> 1) & 7, par = lane & 1, Y = 2 * (lane >> 4) + par;
    const int yb = par ? 1 - (q >> 1) : (q >> 1);
    return (2 * X + (q & 1)) + 16 * (2 * Y + yb);
}
// partner lane of the crossing half of the y-odd colour in that layout (registers 0, 1): the lane of patch Y + 1 for odd Y, Y - 1 for even Y
__host__ __device__ __forceinline__ int sq_patch_ycross(int lane) { return (lane & 1) ? ((lane + 15) & 63) : ((lane + 49) & 63); }

// Partner value of the x-odd colour: odd lanes take lane + 1's v, even lanes lane - 1's (rows of 16, cyclic).  One DPP move
// (row_ror:15) and one v_cndmask with the DPP modifier on its other source (row_ror:1) per word — the compiler's form is two moves
// and a plain select (6 instead of 4 vector-ALU instructions per f64; a fifth of the mat-vec's instructions at 4 slices per wave).
// s_nop 1: a DPP source written by the preceding VALU instruction needs two wait states, which the compiler cannot see in here.
__device__ __forceinline__ double dpp_pair_odd_up(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    int tl, th;
    asm volatile("s_mov_b32 vcc_lo, 0xaaaaaaaa\n\t"
                 "s_mov_b32 vcc_hi, 0xaaaaaaaa\n\t"
                 "s_nop 1\n\t"
                 "v_mov_b32_dpp %0, %2 row_ror:15 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                 "v_mov_b32_dpp %1, %3 row_ror:15 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                 "v_cndmask_b32_dpp %0, %2, %0, vcc row_ror:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                 "v_cndmask_b32_dpp %1, %3, %1, vcc row_ror:1 row_mask:0xf bank_mask:0xf bound_ctrl:0"
                 : "=&v"(tl), "=&v"(th) : "v"(lo), "v"(hi) : "vcc");
    return __hiloint2double(th, tl);
}
// the same for the four values of a slab: the mask and the wait states once
__device__ __forceinline__ void dpp_pair_odd_up4(double (&t)[4], const double (&v)[4]) {
    int a[8], o[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) { a[2 * k] = __double2loint(v[k]); a[2 * k + 1] = __double2hiint(v[k]); }
#define ELPH_DPP_UP "row_ror:15 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
#define ELPH_DPP_DN "row_ror:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
    asm volatile("s_mov_b32 vcc_lo, 0xaaaaaaaa\n\t"
                 "s_mov_b32 vcc_hi, 0xaaaaaaaa\n\t"
                 "s_nop 1\n\t"
                 "v_mov_b32_dpp %0, %8 " ELPH_DPP_UP "v_mov_b32_dpp %1, %9 " ELPH_DPP_UP "v_mov_b32_dpp %2, %10 " ELPH_DPP_UP
                 "v_mov_b32_dpp %3, %11 " ELPH_DPP_UP "v_mov_b32_dpp %4, %12 " ELPH_DPP_UP "v_mov_b32_dpp %5, %13 " ELPH_DPP_UP
                 "v_mov_b32_dpp %6, %14 " ELPH_DPP_UP "v_mov_b32_dpp %7, %15 " ELPH_DPP_UP
                 "v_cndmask_b32_dpp %0, %8, %0, vcc " ELPH_DPP_DN "v_cndmask_b32_dpp %1, %9, %1, vcc " ELPH_DPP_DN
                 "v_cndmask_b32_dpp %2, %10, %2, vcc " ELPH_DPP_DN "v_cndmask_b32_dpp %3, %11, %3, vcc " ELPH_DPP_DN
                 "v_cndmask_b32_dpp %4, %12, %4, vcc " ELPH_DPP_DN "v_cndmask_b32_dpp %5, %13, %5, vcc " ELPH_DPP_DN
                 "v_cndmask_b32_dpp %6, %14, %6, vcc " ELPH_DPP_DN "v_cndmask_b32_dpp %7, %15, %7, vcc row_ror:1 row_mask:0xf bank_mask:0xf bound_ctrl:0"
                 : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7])
                 : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]) : "vcc");
#undef ELPH_DPP_UP
#undef ELPH_DPP_DN
#pragma unroll
    for (int k = 0; k < 4; ++k) t[k] = __hiloint2double(o[2 * k + 1], o[2 * k]);
}

// ---- GRID layout: ANY even-L square lattice, 4 <= L <= 16, in the reference's colouring [x-even | x-odd | y-even | y-odd] ----------------
// (detect_square; cg_wg.hip FORM 5, cg_fast_impl.inc: the Chebyshev recursion).  The 2 x 2 patches of the lattice sit on a G x G grid
// of lanes, G = L / 2 <= 8: lane l < G*G holds the patch X = l % G, Y = l / G; register q is the site x = 2 X + (q & 1),
// y = 2 Y + (q >> 1); site = x + L y.  x-even and y-even bonds pair two registers of a lane; x-odd and y-odd bonds cross to the
// patches X +- 1 / Y +- 1 (cyclically) — by ds_bpermute, whatever G is: the 16 x 16 and 8 x 8 lattices have DPP forms of their own
// (sq_patch_site, s8_site), this is the form of every OTHER size (L = 4, 6, 10, 12, 14): no LDS slab, no per-colour LDS round trip, 8
// crossing values per slab and sweep.  Lanes >= G*G idle: their partners are themselves, they never store and never enter a sum.
// Uniform hopping only: a colour is c (I + th P), the caller applies c^4 (as in the DPP forms).
struct GridCtx {
    double th, k4;           // tanh of the bond angle; c^4
    int xu, xd, yu, yd;      // lanes of the patches X + 1, X - 1, Y + 1, Y - 1
};
// (GX x GY lanes: a rectangular lattice of 2 GX x 2 GY sites, periodic in both directions — the slab of a sharded solve is one: its
//  own rows and ghost rows closed into a ring, shard.hip)
__host__ __device__ __forceinline__ int grid_site(int lane, int q, int GX, int GY) {
    const int l = (lane < GX * GY) ? lane : 0, X = l % GX, Y = l / GX;
    return (2 * X + (q & 1)) + 2 * GX * (2 * Y + (q >> 1));
}
__host__ __device__ __forceinline__ int grid_site(int lane, int q, int G) { return grid_site(lane, q, G, G); }
__device__ __forceinline__ GridCtx grid_ctx(int lane, int GX, int GY, double c, double s) {
    GridCtx X;
    X.th = s / c; X.k4 = (c * c) * (c * c);
    if (lane < GX * GY) {
        const int x = lane % GX, y = lane / GX;
        X.xu = (x + 1) % GX + GX * y; X.xd = (x + GX - 1) % GX + GX * y;
        X.yu = x + GX * ((y + 1) % GY); X.yd = x + GX * ((y + GY - 1) % GY);
    } else {
        X.xu = X.xd = X.yu = X.yd = lane;
    }
    return X;
}
__device__ __forceinline__ GridCtx grid_ctx(int lane, int G, double c, double s) { return grid_ctx(lane, G, G, c, s); }
// one colour on CNT slabs at once: the crossing values of all slabs first (their ds_bpermute round trips overlap), then the arithmetic
template <int CNT, int COL>
__device__ __forceinline__ void grid_colour(double (*v)[4], const GridCtx &X) {
    if constexpr (COL == 0 || COL == 2) {                   // in the lane: x-even (0,1), (2,3); y-even (0,2), (1,3)
        constexpr int a1 = (COL == 0) ? 1 : 2, b0 = (COL == 0) ? 2 : 1;
#pragma unroll
        for (int n = 0; n < CNT; ++n) {
            const double n0 = v[n][0] + X.th * v[n][a1], n1 = v[n][a1] + X.th * v[n][0];
            const double n2 = v[n][b0] + X.th * v[n][3], n3 = v[n][3] + X.th * v[n][b0];
            v[n][0] = n0; v[n][a1] = n1; v[n][b0] = n2; v[n][3] = n3;
        }
    } else {
        // x-odd: the sites x = 2 X + 1 (q = 1, 3) pair with x = 2 X + 2 = q - 1 of the patch X + 1; q = 0, 2 with q + 1 of the patch X - 1
        // y-odd: the sites y = 2 Y + 1 (q = 2, 3) pair with q - 2 of the patch Y + 1; q = 0, 1 with q + 2 of the patch Y - 1
        constexpr int hi0 = (COL == 1) ? 1 : 2, hi1 = 3, lo0 = 0, lo1 = (COL == 1) ? 2 : 1;
        const int up = (COL == 1) ? X.xu : X.yu, dn = (COL == 1) ? X.xd : X.yd;
        double fu0[CNT], fu1[CNT], fd0[CNT], fd1[CNT];
#pragma unroll
        for (int n = 0; n < CNT; ++n) {
            fu0[n] = __shfl(v[n][lo0], up, WAVE); fu1[n] = __shfl(v[n][lo1], up, WAVE);      // the upper neighbour's low-side sites
            fd0[n] = __shfl(v[n][hi0], dn, WAVE); fd1[n] = __shfl(v[n][hi1], dn, WAVE);      // the lower neighbour's high-side sites
        }
#pragma unroll
        for (int n = 0; n < CNT; ++n) {
            v[n][hi0] += X.th * fu0[n]; v[n][hi1] += X.th * fu1[n];
            v[n][lo0] += X.th * fd0[n]; v[n][lo1] += X.th * fd1[n];
        }
    }
    __builtin_amdgcn_sched_barrier(0);          // (a colour of a group of slabs is one scheduling region: hoisted to the front, every slab's crossings spill)
}
// (GRP slabs per scheduling group: 2 — their ds_bpermute round trips overlap; 1 where registers are short: 8 temporaries fewer)
template <int NS, int N0, int COL, int GRP>
__device__ __forceinline__ void grid_pairs(double (&v)[NS][4], const GridCtx &X) {
    grid_colour<(N0 + GRP <= NS) ? GRP : NS - N0, COL>(&v[N0], X);
    if constexpr (N0 + GRP < NS) grid_pairs<NS, N0 + GRP, COL, GRP>(v, X);
}
template <int NS, bool REVERSE, int GRP = 2>
__device__ __forceinline__ void grid_sweepN(double (&v)[NS][4], const GridCtx &X) {
    if constexpr (!REVERSE) { grid_pairs<NS, 0, 0, GRP>(v, X); grid_pairs<NS, 0, 1, GRP>(v, X); grid_pairs<NS, 0, 2, GRP>(v, X); grid_pairs<NS, 0, 3, GRP>(v, X); }
    else                    { grid_pairs<NS, 0, 3, GRP>(v, X); grid_pairs<NS, 0, 2, GRP>(v, X); grid_pairs<NS, 0, 1, GRP>(v, X); grid_pairs<NS, 0, 0, GRP>(v, X); }
}

// ---- HGRID layout: ANY honeycomb lattice of L x L two-site cells (site = 2 (x + L y) + orbital) in the reference's colouring
// [A-B of a cell | B(x,y)-A(x+1,y) | B(x,y)-A(x,y+1)] (detect_honeycomb) whose cells fit a grid of lanes: PX x PY cells per lane
// (NPL = 2 PX PY registers: 1 x 1 -> 2, 2 x 1 -> 4, 2 x 2 -> 8), lanes on a GX x GY grid, GX = L / PX, GY = L / PY, GX GY <= 64 —
// L <= 8 with one cell per lane, L = 10 with two, L = 14 and 16 with four (512 sites in ONE wave).  Register 2 (cx + PX cy) + orbital
// is the site of cell (PX X + cx, PY Y + cy) of lane (X, Y) = (l % GX, l / GX).  A-B pairs registers of the lane; the other two colours
// pair registers of the lane inside the patch and cross to the lane X + 1 / Y + 1 at its edge — by ds_bpermute (the 12 x 12 lattice has a
// DPP form of its own, hc_site).  Uniform hopping: a colour is c (I + th P), the caller applies c^3.
template <int NPL> struct HgDim { static constexpr int PX = (NPL >= 4) ? 2 : 1, PY = (NPL >= 8) ? 2 : 1; };
struct HgCtx {
    double th, k3;
    int xu, xd, yu, yd;      // lanes of the patches X + 1, X - 1, Y + 1, Y - 1 (cyclic); idle lanes: themselves
};
// (LX x LY cells, periodic in both directions; LX = LY for a whole lattice, a sharded solve's slab is LX x rows)
template <int NPL>
__host__ __device__ __forceinline__ int hgrid_site(int lane, int q, int LX, int LY) {
    constexpr int PX = HgDim<NPL>::PX, PY = HgDim<NPL>::PY;
    const int GX = LX / PX, GY = LY / PY;
    const int l = (lane < GX * GY) ? lane : 0, X = l % GX, Y = l / GX;
    const int c = q >> 1, cx = c % PX, cy = c / PX;
    return 2 * ((PX * X + cx) + LX * (PY * Y + cy)) + (q & 1);
}
template <int NPL>
__host__ __device__ __forceinline__ int hgrid_site(int lane, int q, int L) { return hgrid_site<NPL>(lane, q, L, L); }
template <int NPL>
__device__ __forceinline__ HgCtx hgrid_ctx(int lane, int LX, int LY, double c, double s) {
    constexpr int PX = HgDim<NPL>::PX, PY = HgDim<NPL>::PY;
    const int GX = LX / PX, GY = LY / PY;
    HgCtx X;
    X.th = s / c; X.k3 = c * c * c;
    if (lane < GX * GY) {
        const int x = lane % GX, y = lane / GX;
        X.xu = (x + 1) % GX + GX * y; X.xd = (x + GX - 1) % GX + GX * y;
        X.yu = x + GX * ((y + 1) % GY); X.yd = x + GX * ((y + GY - 1) % GY);
    } else {
        X.xu = X.xd = X.yu = X.yd = lane;
    }
    return X;
}
template <int NPL>
__device__ __forceinline__ HgCtx hgrid_ctx(int lane, int L, double c, double s) { return hgrid_ctx<NPL>(lane, L, L, c, s); }
template <int NPL, int CNT, int COL>
__device__ __forceinline__ void hgrid_colour(double (*v)[NPL], const HgCtx &X) {
    constexpr int PX = HgDim<NPL>::PX, PY = HgDim<NPL>::PY;
    auto A = [](int cx, int cy) { return 2 * (cx + PX * cy); };
    auto B = [](int cx, int cy) { return 2 * (cx + PX * cy) + 1; };
    if constexpr (COL == 0) {                              // A-B of a cell
#pragma unroll
        for (int n = 0; n < CNT; ++n)
#pragma unroll
            for (int c = 0; c < PX * PY; ++c) {
                const double a = v[n][2 * c] + X.th * v[n][2 * c + 1], b = v[n][2 * c + 1] + X.th * v[n][2 * c];
                v[n][2 * c] = a; v[n][2 * c + 1] = b;
            }
    } else if constexpr (COL == 1) {                       // B(x,y) - A(x+1,y)
        double fu[CNT][PY], fd[CNT][PY];
#pragma unroll
        for (int n = 0; n < CNT; ++n)
#pragma unroll
            for (int cy = 0; cy < PY; ++cy) {
                fu[n][cy] = __shfl(v[n][A(0, cy)], X.xu, WAVE);             // A(x+1) of my edge B: the next lane's first column
                fd[n][cy] = __shfl(v[n][B(PX - 1, cy)], X.xd, WAVE);        // B(x-1) of my first A: the previous lane's last column
            }
#pragma unroll
        for (int n = 0; n < CNT; ++n)
#pragma unroll
            for (int cy = 0; cy < PY; ++cy) {
                double olda[PX], oldb[PX];
#pragma unroll
                for (int cx = 0; cx < PX; ++cx) { olda[cx] = v[n][A(cx, cy)]; oldb[cx] = v[n][B(cx, cy)]; }
#pragma unroll
                for (int cx = 0; cx < PX; ++cx) {
                    v[n][B(cx, cy)] = oldb[cx] + X.th * ((cx < PX - 1) ? olda[(cx + 1 < PX) ? cx + 1 : 0] : fu[n][cy]);
                    v[n][A(cx, cy)] = olda[cx] + X.th * ((cx > 0) ? oldb[(cx > 0) ? cx - 1 : 0] : fd[n][cy]);
                }
            }
    } else {                                               // B(x,y) - A(x,y+1)
        double fu[CNT][PX], fd[CNT][PX];
#pragma unroll
        for (int n = 0; n < CNT; ++n)
#pragma unroll
            for (int cx = 0; cx < PX; ++cx) {
                fu[n][cx] = __shfl(v[n][A(cx, 0)], X.yu, WAVE);
                fd[n][cx] = __shfl(v[n][B(cx, PY - 1)], X.yd, WAVE);
            }
#pragma unroll
        for (int n = 0; n < CNT; ++n)
#pragma unroll
            for (int cx = 0; cx < PX; ++cx) {
                double olda[PY], oldb[PY];
#pragma unroll
                for (int cy = 0; cy < PY; ++cy) { olda[cy] = v[n][A(cx, cy)]; oldb[cy] = v[n][B(cx, cy)]; }
#pragma unroll
                for (int cy = 0; cy < PY; ++cy) {
                    v[n][B(cx, cy)] = oldb[cy] + X.th * ((cy < PY - 1) ? olda[(cy + 1 < PY) ? cy + 1 : 0] : fu[n][cx]);
                    v[n][A(cx, cy)] = olda[cy] + X.th * ((cy > 0) ? oldb[(cy > 0) ? cy - 1 : 0] : fd[n][cx]);
                }
            }
    }
    __builtin_amdgcn_sched_barrier(0);
}
template <int NPL, int NS, int N0, int COL>
__device__ __forceinline__ void hgrid_pairs(double (&v)[NS][NPL], const HgCtx &X) {
    constexpr int GRP = (NPL >= 8) ? 1 : 2;
    hgrid_colour<NPL, (N0 + GRP <= NS) ? GRP : NS - N0, COL>(&v[N0], X);
    if constexpr (N0 + GRP < NS) hgrid_pairs<NPL, NS, N0 + GRP, COL>(v, X);
}
template <int NPL, int NS, bool REVERSE>
__device__ __forceinline__ void hgrid_sweepN(double (&v)[NS][NPL], const HgCtx &X) {
    if constexpr (!REVERSE) { hgrid_pairs<NPL, NS, 0, 0>(v, X); hgrid_pairs<NPL, NS, 0, 1>(v, X); hgrid_pairs<NPL, NS, 0, 2>(v, X); }
    else                    { hgrid_pairs<NPL, NS, 0, 2>(v, X); hgrid_pairs<NPL, NS, 0, 1>(v, X); hgrid_pairs<NPL, NS, 0, 0>(v, X); }
}

// Honeycomb lattice of 12 x 12 two-site cells, QUAD layout (kpm_sq_dev.h, cg_wg_dev.h): lane 4 y + i (48 of the 64 lanes) holds the
// cells x = 3 i .. 3 i + 2 of lattice row y, register q = 2 b + orbital the site of cell x = 3 i + b; site = 2 (x + 12 y) + orbital.
// Lanes 48..63 shadow lanes 0..15 (valid addresses; they never store and never enter a sum).
__device__ __forceinline__ int hc12q_site(int lane, int q) {
    const int l = (lane < 48) ? lane : lane - 48;
    return 2 * ((3 * (l & 3) + (q >> 1)) + 12 * (l >> 2)) + (q & 1);
}

// wave-wide sum, the same value in every lane: quad swaps and row mirrors (DPP) give every lane the sum of its 16-lane row, the
// four row sums are added in a fixed order through scalar registers.  (The xor butterfly over ds_bpermute it replaces is six
// dependent LDS round trips — ~0.3 us on the critical path of the one-wave kernels, which call it two or three times.)
__device__ __forceinline__ double wave_sum2(double v) {
    v += dpp_f64<0xB1>(v);          // quad_perm [1,0,3,2]
    v += dpp_f64<0x4E>(v);          // quad_perm [2,3,0,1]
    v += dpp_f64<0x141>(v);         // row_half_mirror
    v += dpp_f64<0x140>(v);         // row_mirror
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const double r0 = __hiloint2double(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(lo, 0));
    const double r1 = __hiloint2double(__builtin_amdgcn_readlane(hi, 16), __builtin_amdgcn_readlane(lo, 16));
    const double r2 = __hiloint2double(__builtin_amdgcn_readlane(hi, 32), __builtin_amdgcn_readlane(lo, 32));
    const double r3 = __hiloint2double(__builtin_amdgcn_readlane(hi, 48), __builtin_amdgcn_readlane(lo, 48));
    return (r0 + r1) + (r2 + r3);
}

// sum of n partial sums (one wave; lane l adds entries l, l + 64, ... in that order, then the butterfly).  The loads of a group of
// eight go out together: as a plain loop every coherent load waits for the one before it — ten memory round trips in a row for the
// 640 r.z slots of config C, most of the time k_cg_ap_fast spent for one right-hand side.
__device__ __forceinline__ double reduce_partials_lane(const double *p, int n, int lane) {
    double a = 0.0;
    for (int i0 = lane; i0 < n; i0 += 8 * WAVE) {
        double v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = (i0 + k * WAVE < n) ? ld_coh(p + i0 + k * WAVE) : 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) if (i0 + k * WAVE < n) a += v[k];
    }
    return wave_sum2(a);
}
__device__ __forceinline__ double reduce_partials2(const double *p, int n) { return reduce_partials_lane(p, n, threadIdx.x); }

// Which (right-hand side, chunk) a block of the chunked kernel works on.  order 0: block = rhs * nch + chunk (consecutive blocks — which
// the dispatcher deals round-robin over the 8 XCDs — are the chunks of one right-hand side).  order 1 (XCD-aware; observed placement,
// correctness never depends on it): the blocks of one XCD (equal blockIdx % 8) walk a contiguous range of the order
// [chain][chunk][copy], copy = the right-hand sides that share the chain's exp(-dtau V) (rhs = chain + copy * nchains: the two
// pseudofermion solves of a force evaluation) — so the copies of a chain read the same slices of exp(-dtau V) at the same time on
// the same L2 (one HBM read instead of one per right-hand side: 47 of 691 MB per launch at 144 chains x 2), and tau-neighbouring
// chunks, which re-read each other's boundary slices of p and P^-1 r, follow each other on one XCD.
__device__ __forceinline__ void chunk_block_map(int g, int nrhs, int nch, int nchains, int order, int &rhs, int &ch) {
    const int nb = nrhs * nch;
    if (order == 0 || (nb & 7) || nrhs % nchains) { rhs = g / nch; ch = g - rhs * nch; return; }
    const int per = nrhs / nchains;
    const int s = (g & 7) * (nb >> 3) + (g >> 3);
    const int k = s / (per * nch), rem = s - k * (per * nch);
    ch = rem / per;
    rhs = k + (rem - ch * per) * nchains;
}

// Stop test of an iteration (IterativeSolvers.jl:212-219 / :286-295): eps = |r|/|b| < tol, kappa_min = max_j (2j/ln(2 eps0/eps_j))^2
// > kappa_max, j = maxiter.  Two comparisons screen out the square root, the divisions and the logarithm (~150 dependent f64
// instructions, half a microsecond on the critical path of every wave) while no decision is near: r.r well above (tol |b|)^2
// rules out convergence; (2 eps0/eps)^2 outside [1/2, 2] means |ln(2 eps0/eps)| > 0.34, which rules out the kappa stop while
// 2j < 0.34 sqrt(kappa_max).  Near a decision, on the last iteration and with a residual history the reference's arithmetic runs
// (eps, and kappa_min of that iteration: an iteration stops on kappa only through its own term).  Returns the done code.
__device__ __forceinline__ int cg_stop_test(const CgParams &P, const CgState &S, double rr, long long seq, double &eps, double &kmin) {
    const double tb = P.tol * S.normb, y_num = 4.0 * (S.eps0 * S.normb) * (S.eps0 * S.normb);
    const bool screened = !P.record_hist && seq < P.maxiter && rr > tb * tb * 1.000001 &&
                          (rr + rr <= y_num || rr >= y_num + y_num) && (double)seq < 0.17 * sqrt(P.kmax);
    if (screened) return 0;
    eps = sqrt(rr) / S.normb;
    const double qq = 2.0 * (double)seq / log(2.0 * S.eps0 / eps);
    const double val = qq * qq;
    kmin = (val > kmin) ? val : kmin;
    if (eps < P.tol) return 1;
    if (kmin > P.kmax) return 2;
    if (seq >= P.maxiter) return 3;
    return 0;
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
