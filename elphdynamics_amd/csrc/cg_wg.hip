// cg_wg.hip — workgroup-resident conjugate gradient: the whole un-preconditioned solve of M^T M x = b
// (IterativeSolvers.jl:239-314) in ONE launch, Krylov vectors in registers, for lattices with a 4-colour lane program.
//
// Why: the two-kernel iteration (cg_fast_impl.inc) streams r, p, x, z through HBM/L2 every iteration — HBM-bound in a large
// batch (0.72 of the 8 TB/s peak), launch-bound for the reference's real call shape (1-2 right-hand sides: 10 us per
// iteration for 0.6 us of work).  Here a right-hand side belongs to a TEAM of G workgroups of W wavefronts; a wavefront owns
// T consecutive tau-slices for the whole solve and keeps x, r, p (own slices + one halo slice each side) and exp(-dtau V)
// in registers.  Nothing of the Krylov vectors goes back to memory between iterations: an iteration costs two
// checkerboard-sweep passes in the wave's private LDS slabs plus two MEETINGS of the team:
//   (1) p.z  ->  alpha            (2) r.r + the boundary slices of the new r  ->  eps, stop test, beta, halo of the new p
// A meeting is hierarchical: wave partials -> LDS -> s_barrier (inside the workgroup), then — only if the team has more
// than one workgroup — one 16-byte record per workgroup through L2: two 8-byte {tag = iteration, half of the f64} granules
// written by ONE sc1 store each and polled by every wave with sc1 loads until all tags carry the iteration number
// (cdna_hip_programming.md, Guideline 16, form R2: the data is the flag; no fence).  Boundary slices of r (2 KB) that cross
// a workgroup boundary are sc1-stored before the workgroup's barrier and the r.r record that follows it is their flag
// (form R1: every storing wave drains vmcnt before the barrier; consumers read them with sc1 loads after their own poll).
// Every wave of a team reduces the same records in the same order, so alpha, beta and the stop decision are bit-identical
// across the team and from run to run.
//
// Placement: team members are blocks with equal blockIdx % 8 (they share an XCD and its L2 under the observed round-robin
// placement — speed only, never correctness).  The grid may hold more teams than the chip can keep resident: blocks are
// dispatched in index order, a team's members have neighbouring indices, so at most the eight teams at the dispatch
// frontier wait (spinning) for members that start when another team has finished its solve.  Every spin is bounded by a
// wall-clock limit; a wave that gives up raises `abort`, all others leave within 64 polls, and the host falls back to the
// two-kernel iteration.

#include <algorithm>

#include "cg_fast_common.h"

namespace wg {

constexpr int MC = 4;                       // colours of the lane program (square / honeycomb / chain lattices)
typedef unsigned long long u64;

template <int NPL>
__host__ __device__ constexpr int slab_len() { return NPL * WAVE + 2 * WAVE; }

struct WgCtl {
    u64 *slots;          // [nrhs][2 meetings][Gmax = 32][2 granules]; zeroed by the host before every launch
    double *bnd;         // [nrhs][G][2][NPL*64]: first / last slice of r of every workgroup (G > 1 only)
    int *abort;
    int G, W;
    long long timeout_ticks;   // wall_clock64 ticks (100 MHz)
    long long fixed_iters;     // > 0: measurement mode, exactly this many iterations, no stop test
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

// forward checkerboard sweep on two independent slabs (Checkerboard.jl:57-83), bonds in registers
template <int NPL, bool UNI>
__device__ __forceinline__ void sweep2(double *bufA, double *bufB, const unsigned (&ij)[MC * ((NPL + 1) / 2)],
                                       const Tab<MC * ((NPL + 1) / 2), UNI> &tA, const Tab<MC * ((NPL + 1) / 2), UNI> &tB, int ncol) {
    constexpr int PP = (NPL + 1) / 2;
#pragma unroll
    for (int col = 0; col < MC; ++col) {
        if (col < ncol) {
            double a0[PP], a1[PP], b0[PP], b1[PP];
#pragma unroll
            for (int pp = 0; pp < PP; ++pp) {
                const unsigned w = ij[col * PP + pp];
                a0[pp] = bufA[w & 0xFFFF]; a1[pp] = bufA[w >> 16];
                b0[pp] = bufB[w & 0xFFFF]; b1[pp] = bufB[w >> 16];
            }
#pragma unroll
            for (int pp = 0; pp < PP; ++pp) {
                const int e = col * PP + pp;
                const unsigned w = ij[e];
                bufA[w & 0xFFFF] = tA.C(e) * a0[pp] + tA.S(e) * a1[pp];
                bufA[w >> 16] = tA.C(e) * a1[pp] + tA.S(e) * a0[pp];
                bufB[w & 0xFFFF] = tB.C(e) * b0[pp] + tB.S(e) * b1[pp];
                bufB[w >> 16] = tB.C(e) * b1[pp] + tB.S(e) * b0[pp];
            }
            WAVE_LDS_ORDER();
        }
    }
}

// forward sweep on bufA (tables A) and reverse sweep on bufB (tables B) in the same four colour stages
template <int NPL, bool UNI>
__device__ __forceinline__ void sweep_fr(double *bufA, double *bufB, const unsigned (&ij)[MC * ((NPL + 1) / 2)],
                                         const Tab<MC * ((NPL + 1) / 2), UNI> &tA, const Tab<MC * ((NPL + 1) / 2), UNI> &tB, int ncol,
                                         bool doA) {
    constexpr int PP = (NPL + 1) / 2;
#pragma unroll
    for (int cc = 0; cc < MC; ++cc) {
        const int colA = cc, colB = MC - 1 - cc;
        const bool onA = doA && (colA < ncol), onB = (colB < ncol);
        if (onA || onB) {
            double a0[PP], a1[PP], b0[PP], b1[PP];
#pragma unroll
            for (int pp = 0; pp < PP; ++pp) {
                if (onA) { const unsigned w = ij[colA * PP + pp]; a0[pp] = bufA[w & 0xFFFF]; a1[pp] = bufA[w >> 16]; }
                if (onB) { const unsigned w = ij[colB * PP + pp]; b0[pp] = bufB[w & 0xFFFF]; b1[pp] = bufB[w >> 16]; }
            }
#pragma unroll
            for (int pp = 0; pp < PP; ++pp) {
                if (onA) {
                    const int e = colA * PP + pp; const unsigned w = ij[e];
                    bufA[w & 0xFFFF] = tA.C(e) * a0[pp] + tA.S(e) * a1[pp];
                    bufA[w >> 16] = tA.C(e) * a1[pp] + tA.S(e) * a0[pp];
                }
                if (onB) {
                    const int e = colB * PP + pp; const unsigned w = ij[e];
                    bufB[w & 0xFFFF] = tB.C(e) * b0[pp] + tB.S(e) * b1[pp];
                    bufB[w >> 16] = tB.C(e) * b1[pp] + tB.S(e) * b0[pp];
                }
            }
            WAVE_LDS_ORDER();
        }
    }
}

__device__ __forceinline__ void st_gran(u64 *p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u64 ld_gran(const u64 *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_sc1(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld_sc1(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Team meeting through L2.  Called by EVERY wave of every workgroup of the team with the workgroup's value `mine` (identical
// in all its waves); the `publisher` wave stores the workgroup's record.  On return every wave holds the sum of the G
// records taken in index order.  false: timed out or aborted.
__device__ __forceinline__ bool team_sum(u64 *slots, int g, int G, double mine, unsigned epoch, bool publisher, int lane,
                                         const WgCtl &R, double &total) {
    if (publisher && lane < 2) {
        const u64 bits = (u64)__double_as_longlong(mine);
        const unsigned half = lane ? (unsigned)(bits >> 32) : (unsigned)bits;
        st_gran(slots + 2 * g + lane, ((u64)epoch << 32) | half);
    }
    u64 v = 0;
    long long t_start = 0;
    for (int spin = 0;; ++spin) {
        bool ok = true;
        if (lane < 2 * G) { v = ld_gran(slots + lane); ok = ((unsigned)(v >> 32) == epoch); }
        if (__all(ok)) break;
        if ((spin & 63) == 63) {
            if (__hip_atomic_load(R.abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
            const long long now = wall_clock64();
            if (t_start == 0) t_start = now;
            else if (now - t_start > R.timeout_ticks) {
                if (lane == 0) __hip_atomic_store(R.abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return false;
            }
        }
        __builtin_amdgcn_s_sleep(1);
    }
    const unsigned lo = (unsigned)v;
    const unsigned hi = (unsigned)__shfl((int)lo, lane | 1, WAVE);             // even lane 2k: the high half sits in lane 2k+1
    const double val = __longlong_as_double((long long)(((u64)hi << 32) | lo));
    total = 0.0;
    for (int k = 0; k < G; ++k) total += __shfl(val, 2 * k, WAVE);
    return true;
}

template <int NPL, int T, bool SSH, bool UNI>
__global__ void __launch_bounds__(512) k_cg_wg(CgBufs B, ModelDev m, WgCtl R) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int NE = MC * ((NPL + 1) / 2);
    constexpr int HS = NPL * WAVE, SL = slab_len<NPL>();
    constexpr int NH = (T == 1) ? 1 : 2;               // boundary slices a wave shows its neighbours (T = 1: first == last)
    constexpr int NT = SSH ? T + 1 : 1;                // hopping-table sets (SSH: one per slice t0 .. t0+T)
    constexpr int NEJ = SSH ? 1 : T + 1;               // exp(-dtau V) slices (SSH: exp(dtau mu), per site only)
    const int W = R.W, G = R.G;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & (WAVE - 1);
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int tq = idx / G, g = idx - tq * G;
    const int rhs = tq * 8 + xcd;
    if (rhs >= B.nrhs) return;
    const int N = m.N, L = m.L;
    const int t0 = (g * W + wv) * T;
    const size_t ndim = (size_t)N * L;
    double *bufA = lds + (size_t)wv * 2 * SL, *bufB = bufA + SL;
    double *hal = lds + (size_t)W * 2 * SL;            // [W][NH][HS]
    double *partA = hal + (size_t)W * NH * HS, *partB = partA + W;
    auto wrap = [L](int t) { return (t < 0) ? t + L : ((t >= L) ? t - L : t); };
    auto sgn = [](int t) { return (t == 0) ? -1.0 : 1.0; };
    const CgParams P = B.params;
    CgState *st2 = B.state + 2 * rhs;
    const CgState S = ld_state(st2);
    if (S.done || S.seq != 0) return;                  // fresh solves only (the host guarantees it)

    int sc[NPL];
#pragma unroll
    for (int q = 0; q < NPL; ++q) { const int s = lane + q * WAVE; sc[q] = (s < N) ? s : N - 1; }
    double *xg = B.x + (size_t)rhs * ndim, *rg = B.r + (size_t)rhs * ndim;
    const double *pg = B.p + (size_t)rhs * ndim;       // parity 0: p0 of k_cg_init
    const double *Ech = m.E + (size_t)(rhs % m.nchains) * m.E_chain_stride;

    // x lives in memory: it is only ever updated (x += alpha p), never an input of the iteration — its load-add-store rides under
    // the second meeting and costs no register across the mat-vec
    double r[T][NPL], z[T][NPL], p[T + 2][NPL], E[NEJ][NPL];
#pragma unroll
    for (int j = 0; j < T; ++j)
#pragma unroll
        for (int q = 0; q < NPL; ++q) r[j][q] = rg[(size_t)(t0 + j) * N + sc[q]];
#pragma unroll
    for (int j = 0; j < T + 2; ++j)
#pragma unroll
        for (int q = 0; q < NPL; ++q) p[j][q] = pg[(size_t)wrap(t0 + j - 1) * N + sc[q]];
#pragma unroll
    for (int j = 0; j < NEJ; ++j)
#pragma unroll
        for (int q = 0; q < NPL; ++q) E[j][q] = Ech[(size_t)wrap(t0 + j) * m.E_tau_stride + sc[q]];
    unsigned ij[NE];
    Tab<NE, UNI> tab[NT];
    load_ij<NPL>(ij, m, lane);
    if (SSH) {
        ssh_chain_select(m, rhs);
#pragma unroll
        for (int j = 0; j < NT; ++j)
            load_tab<NE, UNI>(tab[j], m.lp_c + (size_t)wrap(t0 + j) * m.lp_tau_stride, m.lp_s + (size_t)wrap(t0 + j) * m.lp_tau_stride, lane, m);
    } else {
        load_tab<NE, UNI>(tab[0], m.lp_c, m.lp_s, lane, m);
    }
#define TAB(j) tab[SSH ? (j) : 0]
#define EXPV(j) E[SSH ? 0 : (j)]

    u64 *slotsA = R.slots + (size_t)rhs * 2 * 64, *slotsB = slotsA + 64;
    double *bnd = R.bnd + (size_t)rhs * G * 2 * HS;
    const int gm = (g == 0) ? G - 1 : g - 1, gp = (g == G - 1) ? 0 : g + 1;
    double rho = S.rho, kmin = S.kmin, eps = S.eps;
    const double eps0 = S.eps0, normb = S.normb;

    for (long long seq = 0;; ++seq) {
        const unsigned epoch = (unsigned)seq + 1u;
        // ---- z = M^T M p on the own slices: w(t) = p(t) - sg(t) CB_t [E(t) p(t-1)],  z(t) = w(t) - sg(t+1) E(t+1) CB_{t+1}^T w(t+1)
        double wprev[NPL], wcur[NPL];
#pragma unroll
        for (int q = 0; q < NPL; ++q) {
            const int s = lane + q * WAVE;
            bufA[s] = EXPV(0)[q] * p[0][q];
            bufB[s] = EXPV(1)[q] * p[1][q];
        }
        WAVE_LDS_ORDER();
        sweep2<NPL, UNI>(bufA, bufB, ij, TAB(0), TAB(1), m.ncol);
        {
            const double sga = sgn(t0), sgb = sgn(wrap(t0 + 1));
#pragma unroll
            for (int q = 0; q < NPL; ++q) {
                const int s = lane + q * WAVE;
                wprev[q] = p[1][q] - sga * bufA[s];
                wcur[q] = p[2][q] - sgb * bufB[s];
            }
        }
        WAVE_LDS_ORDER();
        double acc = 0.0;
#pragma unroll
        for (int j = 1; j <= T; ++j) {
            const bool more = (j < T);
#pragma unroll
            for (int q = 0; q < NPL; ++q) {
                const int s = lane + q * WAVE;
                bufB[s] = wcur[q];
                if (more) bufA[s] = EXPV((j + 1 <= T) ? j + 1 : T)[q] * p[j + 1][q];        // E(t0+j+1) .* p(t0+j)
            }
            WAVE_LDS_ORDER();
            sweep_fr<NPL, UNI>(bufA, bufB, ij, TAB((j + 1 <= T) ? j + 1 : T), TAB(j), m.ncol, more);
            const double sgj = sgn(wrap(t0 + j)), sgnn = sgn(wrap(t0 + j + 1));
            double wnext[NPL];
#pragma unroll
            for (int q = 0; q < NPL; ++q) {
                const int s = lane + q * WAVE;
                const double zz = wprev[q] - sgj * EXPV(j)[q] * bufB[s];                      // z(t0+j-1)
                z[j - 1][q] = zz;
                if (s < N) acc += p[j][q] * zz;
                if (more) wnext[q] = p[(j + 2 <= T + 1) ? j + 2 : T + 1][q] - sgnn * bufA[s];  // w(t0+j+1)
            }
            WAVE_LDS_ORDER();
            if (more) {
#pragma unroll
                for (int q = 0; q < NPL; ++q) { wprev[q] = wcur[q]; wcur[q] = wnext[q]; }
            }
        }
        acc = wave_sum2(acc);
        // ---- meeting 1: p.z ---------------------------------------------------------------------------------------
        if (lane == 0) partA[wv] = acc;
        __syncthreads();
        double pap = 0.0;
        for (int i = 0; i < W; ++i) pap += partA[i];
        if (G > 1) {
            double tot;
            if (!team_sum(slotsA, g, G, pap, epoch, wv == 0, lane, R, tot)) return;
            pap = tot;
        }
        const double alpha = rho / pap;
        // ---- r -= alpha z, r.r; show the boundary slices of the new r; x += alpha p goes to memory under the second meeting ---
        double a = 0.0;
#pragma unroll
        for (int j = 0; j < T; ++j)
#pragma unroll
            for (int q = 0; q < NPL; ++q) {
                const int s = lane + q * WAVE;
                const double rn = r[j][q] - alpha * z[j][q];
                r[j][q] = rn;
                if (s < N) a += rn * rn;
            }
        a = wave_sum2(a);
#pragma unroll
        for (int q = 0; q < NPL; ++q) {
            const int s = lane + q * WAVE;
            hal[((size_t)wv * NH + 0) * HS + s] = r[0][q];
            if (NH == 2) hal[((size_t)wv * NH + 1) * HS + s] = r[T - 1][q];
        }
        if (G > 1) {
            if (wv == 0) {
#pragma unroll
                for (int q = 0; q < NPL; ++q) st_sc1(bnd + ((size_t)g * 2 + 0) * HS + lane + q * WAVE, r[0][q]);
            }
            if (wv == W - 1) {
#pragma unroll
                for (int q = 0; q < NPL; ++q) st_sc1(bnd + ((size_t)g * 2 + 1) * HS + lane + q * WAVE, r[T - 1][q]);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wave drains before the barrier; the record after it is the flag
        }
        if (lane == 0) partB[wv] = a;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < T; ++j)
#pragma unroll
            for (int q = 0; q < NPL; ++q) {
                const int s = lane + q * WAVE;
                if (s < N) { const size_t i = (size_t)(t0 + j) * N + s; xg[i] = xg[i] + alpha * p[j + 1][q]; }     // IterativeSolvers.jl:282
            }
        double rr = 0.0;
        for (int i = 0; i < W; ++i) rr += partB[i];
        if (G > 1) {
            double tot;
            if (!team_sum(slotsB, g, G, rr, epoch, wv == 0, lane, R, tot)) return;
            rr = tot;
        }
        // ---- stop test of iteration it = seq + 1 (IterativeSolvers.jl:286-295) ---------------------------------------------
        const long long it = seq + 1;
        int done = 0;
        if (R.fixed_iters > 0) {
            if (it >= R.fixed_iters) done = 3;
        } else {
            eps = sqrt(rr) / normb;
            const double qq = 2.0 * (double)it / log(2.0 * eps0 / eps);
            const double val = qq * qq;
            kmin = (val > kmin) ? val : kmin;
            if (eps < P.tol) done = 1;
            else if (kmin > P.kmax) done = 2;
            else if (it >= P.maxiter) done = 3;
            if (g == 0 && wv == 0 && lane == 0 && P.record_hist) B.hist[(size_t)rhs * P.hist_stride + it] = eps;
        }
        if (done) {
#pragma unroll
            for (int j = 0; j < T; ++j)
#pragma unroll
                for (int q = 0; q < NPL; ++q) {
                    const int s = lane + q * WAVE;
                    if (s < N) rg[(size_t)(t0 + j) * N + s] = r[j][q];
                }
            if (g == 0 && wv == 0 && lane == 0) {
                CgState o = S;
                o.rho = rho; o.kmin = kmin; o.eps = eps; o.seq = it + 1; o.iters = it; o.done = done;
                st2[0] = o;
                st2[1] = o;
            }
            return;
        }
        const double beta = rr / rho;
        rho = rr;
        // ---- next direction on the own slices and on the two halo slices (p = r + beta p is pointwise) --------------------
        double hl[NPL], hr[NPL];
        if (wv > 0 || G == 1) {
            const double *src = hal + ((size_t)((wv > 0) ? wv - 1 : W - 1) * NH + (NH - 1)) * HS;
#pragma unroll
            for (int q = 0; q < NPL; ++q) hl[q] = src[lane + q * WAVE];
        } else {
            const double *src = bnd + ((size_t)gm * 2 + 1) * HS;
#pragma unroll
            for (int q = 0; q < NPL; ++q) hl[q] = ld_sc1(src + lane + q * WAVE);
        }
        if (wv < W - 1 || G == 1) {
            const double *src = hal + ((size_t)((wv < W - 1) ? wv + 1 : 0) * NH + 0) * HS;
#pragma unroll
            for (int q = 0; q < NPL; ++q) hr[q] = src[lane + q * WAVE];
        } else {
            const double *src = bnd + ((size_t)gp * 2 + 0) * HS;
#pragma unroll
            for (int q = 0; q < NPL; ++q) hr[q] = ld_sc1(src + lane + q * WAVE);
        }
#pragma unroll
        for (int q = 0; q < NPL; ++q) {
            p[0][q] = hl[q] + beta * p[0][q];
            p[T + 1][q] = hr[q] + beta * p[T + 1][q];
        }
#pragma unroll
        for (int j = 0; j < T; ++j)
#pragma unroll
            for (int q = 0; q < NPL; ++q) p[j + 1][q] = r[j][q] + beta * p[j + 1][q];
    }
#undef TAB
#undef EXPV
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------

struct Shape { int T, W, G; size_t shm; };

// T = 2 slices per wave where the register file takes it (site phonons, <= 4 sites per lane: no spills at 256 VGPRs), else 1;
// W = the largest divisor of Ltau / T that is <= 8 waves (two per SIMD), G = workgroups per right-hand side
static bool pick_shape(const elph_handle_s *h, int forceT, Shape *out) {
    const int L = (int)h->L;
    const bool ssh = (h->kind == ELPH_MODEL_SSH);
    const int cand[2] = {2, 1};
    for (int T : cand) {
        if (forceT && T != forceT) continue;
        if (T == 2 && (ssh || h->npl > 4)) continue;
        if (L % T) continue;
        const int Wt = L / T;
        int W = 0;
        for (int w = std::min(8, Wt); w >= 1; --w) if (Wt % w == 0) { W = w; break; }
        const int G = Wt / W;
        if (G > 32) continue;                            // the 2G record granules of a meeting must fit one wave's poll
        const size_t SL = (size_t)h->npl * WAVE + 2 * WAVE, HS = (size_t)h->npl * WAVE, NH = (T == 1) ? 1 : 2;
        const size_t shm = ((size_t)W * 2 * SL + (size_t)W * NH * HS + 2 * (size_t)W + 2) * sizeof(double);
        if (shm > 160 * 1024) continue;
        out->T = T; out->W = W; out->G = G; out->shm = shm;
        return true;
    }
    return false;
}

template <int NPL, int T, bool SSH, bool UNI>
static hipError_t launch_k(elph_handle_s *h, const Shape &sh, dim3 grid, const CgBufs &B, const ModelDev &m, const WgCtl &R) {
    hipError_t e = hipFuncSetAttribute((const void *)k_cg_wg<NPL, T, SSH, UNI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh.shm);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_cg_wg<NPL, T, SSH, UNI>), grid, dim3(sh.W * WAVE), sh.shm, h->stream, B, m, R);
    return hipGetLastError();
}

template <int NPL>
static hipError_t launch_npl(elph_handle_s *h, const Shape &sh, dim3 grid, const CgBufs &B, const ModelDev &m, const WgCtl &R) {
    if (h->kind == ELPH_MODEL_SSH) return launch_k<NPL, 1, true, false>(h, sh, grid, B, m, R);
    if constexpr (NPL <= 4) {
        if (sh.T == 2) return m.uniform ? launch_k<NPL, 2, false, true>(h, sh, grid, B, m, R) : launch_k<NPL, 2, false, false>(h, sh, grid, B, m, R);
    }
    return m.uniform ? launch_k<NPL, 1, false, true>(h, sh, grid, B, m, R) : launch_k<NPL, 1, false, false>(h, sh, grid, B, m, R);
}

}  // namespace wg

// Whether the workgroup-resident kernel can run this handle's un-preconditioned solves (and with which shape).
bool elph_wg_usable(const elph_handle_s *h, int *T, int *W, int *G) {
    static const bool off = []() { const char *e = getenv("ELPH_NO_WG"); return e && e[0] == '1'; }();
    if (off || !h->fast || h->lp_mc != 4 || h->npl > 5 || h->dot_hi != 0 || h->solo_chain >= 0) return false;
    const char *et = getenv("ELPH_WG_T");
    wg::Shape sh;
    if (!wg::pick_shape(h, et ? atoi(et) : 0, &sh)) return false;
    if (T) *T = sh.T;
    if (W) *W = sh.W;
    if (G) *G = sh.G;
    return true;
}

// Runs the whole un-preconditioned CG for rhs [0, nrhs) after elph_launch_cg_init (fixed_iters > 0: exactly that many
// iterations without stop test — measurement).  *ran = false: not applicable, nothing was launched.
// ELPH_E_HIP with "workgroup-resident" in the message: a team timed out; the caller re-initialises and runs the two-kernel path.
int elph_wg_cg(elph_handle_s *h, const CgBufs &B, int nrhs, long long fixed_iters, bool *ran) {
    *ran = false;
    if (h->wg_broken || B.params.use_prec) return ELPH_OK;
    if (!elph_wg_usable(h, nullptr, nullptr, nullptr)) return ELPH_OK;
    const char *et = getenv("ELPH_WG_T");
    wg::Shape sh;
    if (!wg::pick_shape(h, et ? atoi(et) : 0, &sh)) return ELPH_OK;
    const size_t HS = (size_t)h->npl * WAVE;
    const size_t n_slots = (size_t)nrhs * 2 * 64, n_bnd = (size_t)nrhs * sh.G * 2 * HS;
    const size_t need = n_slots * sizeof(wg::u64) + 64 + n_bnd * sizeof(double);
    if (need > h->res_cap) {
        HIPCHK(hipStreamSynchronize(h->stream));
        if (h->d_res) HIPCHK(hipFree(h->d_res));
        h->d_res = nullptr;
        HIPCHK(hipMalloc(&h->d_res, need));
        h->res_cap = need;
    }
    wg::WgCtl R;
    char *base = static_cast<char *>(h->d_res);
    R.slots = reinterpret_cast<wg::u64 *>(base);
    R.abort = reinterpret_cast<int *>(base + n_slots * sizeof(wg::u64));
    R.bnd = reinterpret_cast<double *>(base + n_slots * sizeof(wg::u64) + 64);
    R.G = sh.G; R.W = sh.W;
    const char *eto = getenv("ELPH_WG_TIMEOUT_MS");
    R.timeout_ticks = (long long)(eto ? atoll(eto) : 20000) * 100000LL;     // wall_clock64 runs at 100 MHz
    R.fixed_iters = fixed_iters;
    HIPCHK(hipMemsetAsync(base, 0, n_slots * sizeof(wg::u64) + 64, h->stream));   // every polled word, every launch
    ModelDev m = elph_model_dev(h);
    const dim3 grid((unsigned)(8 * ((nrhs + 7) / 8) * sh.G));
    hipError_t e = hipSuccess;
    switch (h->npl) {
        case 1: e = wg::launch_npl<1>(h, sh, grid, B, m, R); break;
        case 2: e = wg::launch_npl<2>(h, sh, grid, B, m, R); break;
        case 3: e = wg::launch_npl<3>(h, sh, grid, B, m, R); break;
        case 4: e = wg::launch_npl<4>(h, sh, grid, B, m, R); break;
        default: e = wg::launch_npl<5>(h, sh, grid, B, m, R); break;
    }
    if (e != hipSuccess) { elph_set_error("launch k_cg_wg failed: %s", hipGetErrorString(e)); return ELPH_E_HIP; }
    h->wg_T = sh.T; h->wg_W = sh.W; h->wg_G = sh.G;
    h->wg_abort_off = n_slots * sizeof(wg::u64);
    *ran = true;
    return ELPH_OK;
}

// after the stream has drained: did a team give up?  (the abort word follows the records in the control block)
int elph_wg_aborted(elph_handle_s *h, bool *aborted) {
    *aborted = false;
    if (!h->d_res || h->wg_abort_off == 0) return ELPH_OK;
    int ab = 0;
    HIPCHK(hipMemcpy(&ab, static_cast<char *>(h->d_res) + h->wg_abort_off, sizeof(int), hipMemcpyDeviceToHost));
    if (ab) {
        h->wg_broken = true;
        *aborted = true;
        elph_set_error("workgroup-resident CG timed out waiting for its team (T=%d W=%d G=%d); falling back to the two-kernel iteration",
                       h->wg_T, h->wg_W, h->wg_G);
    }
    return ELPH_OK;
}
