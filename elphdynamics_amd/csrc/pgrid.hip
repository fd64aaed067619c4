// pgrid.hip — register-exchange kernels for even-L square lattices beyond 16 x 16 (L = 18, 20, 24, 28, 32; pgrid_dev.h: a PX x PY
// patch of sites per lane, the whole time slice in ONE wavefront), uniform hopping, site phonons (Holstein).
//
// k_kpm_cheb_pg: the per-frequency Chebyshev recursion of the KPM preconditioner (KPMPreconditioners.jl:426-481 ldiv!, :606-693 the
// series) — what the generic kernel (kernels.hip: k_kpm_cheb, a workgroup of up to 1024 threads per (right-hand side, frequency), one
// LDS round trip and two barriers per checkerboard colour) spends 180-216 us on for ONE right-hand side at L = 24 / 32.  Here a block
// is two wavefronts — the real and the imaginary parts of nu_w — and a step of the recursion is the patch sweep (4 colours, no
// barrier) plus four instructions per site; the two waves meet through LDS only where the complex coefficients mix them (twice per
// frequency).
#include <algorithm>
#include <cstdlib>

#include "elph_internal.h"
#include "pgrid_dev.h"

#define WAVE ELPH_WAVE

namespace {

using SQ44 = pgrid::Sq<4, 4>; using SQ26 = pgrid::Sq<2, 6>; using SQ24 = pgrid::Sq<2, 4>; using SQ2A = pgrid::Sq<2, 10>; using SQ46 = pgrid::Sq<4, 6>;
// several wavefronts per slice (round 6): 2 x 2 patches on 2, 3, 5, 6 wavefronts (L = 22, 26, 34, 38), 4 x 4 patches on 2, 3, 4 (L = 40 ... 64)
using M22_2 = pgrid::Sq<2, 2, 2>; using M22_3 = pgrid::Sq<2, 2, 3>; using M22_4 = pgrid::Sq<2, 2, 4>; using M22_5 = pgrid::Sq<2, 2, 5>; using M22_6 = pgrid::Sq<2, 2, 6>;
using M44_2 = pgrid::Sq<4, 4, 2>; using M44_3 = pgrid::Sq<4, 4, 3>; using M44_4 = pgrid::Sq<4, 4, 4>;
// hopping disorder (round 6): a (cosh, sinh) pair per bond from a table in LDS — the shapes of pgrid::patch_takes_disorder
using SQ44D = pgrid::Sq<4, 4, 1, false>; using SQ26D = pgrid::Sq<2, 6, 1, false>; using SQ24D = pgrid::Sq<2, 4, 1, false>;
using M22_2D = pgrid::Sq<2, 2, 2, false>; using M22_3D = pgrid::Sq<2, 2, 3, false>; using M22_4D = pgrid::Sq<2, 2, 4, false>; using M22_5D = pgrid::Sq<2, 2, 5, false>;
using H33_2 = pgrid::Hc<3, 3, 2>; using H33_3 = pgrid::Hc<3, 3, 3>; using H33_4 = pgrid::Hc<3, 3, 4>;      // honeycomb: 3 x 3 / 2 x 2 cells per thread
using H22_2 = pgrid::Hc<2, 2, 2>; using H22_3 = pgrid::Hc<2, 2, 3>; using H22_4 = pgrid::Hc<2, 2, 4>;
using TR22 = pgrid::Tri<2, 2>; using TR24 = pgrid::Tri<2, 4>; using TR26 = pgrid::Tri<2, 6>; using TR44 = pgrid::Tri<4, 4>;
using HC32 = pgrid::Hc<3, 2>; using HC42 = pgrid::Hc<4, 2>; using HC33 = pgrid::Hc<3, 3>;

__device__ __forceinline__ double pg_wave_sum(double v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}
// the sum over the NW wavefronts of a slice (threads t0 ... t0 + 64 NW - 1 of the block; `red`: NW doubles of LDS of that group): wave sums, added
// in wave order — the same value in every thread.  One wavefront: the wave sum.  (Every thread of the group must call it: two barriers.)
template <int NW>
__device__ __forceinline__ double pg_group_sum(double v, double *red, int ltid) {
    v = pg_wave_sum(v);
    if constexpr (NW == 1) return v;
    else {
        __syncthreads();
        if ((ltid & (WAVE - 1)) == 0) red[ltid >> 6] = v;
        __syncthreads();
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) t += red[w];
        return t;
    }
}

// hopping disorder: this thread's (cosh, sinh) pairs -> tab (LDS; pgrid::Ctx::tab), from the bond tables through the site -> bond map of the
// colouring (pgb[colour][site]); every thread reads back only what it wrote (the two parts of the Chebyshev kernel write the same values to the
// same words), so no barrier is needed
extern __shared__ __attribute__((aligned(16))) double2 pg_dtab[];
template <class LAT, int COL>
__device__ __forceinline__ void pg_fill_col(int lane, int Ls, int N, const int *__restrict__ pgb, const double *__restrict__ cb, const double *__restrict__ sb) {
    using D = pgrid::ColDims<LAT::PX, LAT::PY, COL>;
    constexpr int NT = LAT::NW * WAVE, BASE = pgrid::col_base<LAT::PX, LAT::PY, COL>();
    auto put = [&](int slot, int reg) {
        const int b = pgb[(size_t)COL * N + LAT::site_of(lane, reg, Ls)];
        pg_dtab[(size_t)(BASE + slot) * NT + lane] = make_double2(cb[b], sb[b]);
    };
#pragma unroll
    for (int b = 0; b < D::PB; ++b) {
#pragma unroll
        for (int a = D::ODD ? 1 : 0; a + 1 < D::PA; a += 2) put(D::pair_slot(a, b), a * D::SA + b * D::SB);
        if constexpr (D::ODD) {
            put(D::edge_slot(true, b), (D::PA - 1) * D::SA + b * D::SB);
            put(D::edge_slot(false, b), 0 * D::SA + b * D::SB);
        }
    }
}
template <class LAT>
__device__ __forceinline__ void pg_fill_tab(pgrid::Ctx &X, int lane, int Ls, int N, const int *__restrict__ pgb, const double *__restrict__ cb,
                                            const double *__restrict__ sb) {
    if constexpr (LAT::DIS) {
        pg_fill_col<LAT, 0>(lane, Ls, N, pgb, cb, sb);
        pg_fill_col<LAT, 1>(lane, Ls, N, pgb, cb, sb);
        pg_fill_col<LAT, 2>(lane, Ls, N, pgb, cb, sb);
        pg_fill_col<LAT, 3>(lane, Ls, N, pgb, cb, sb);
        X.tab = pg_dtab;
    }
}

// sum_n c_n T_n(A') v  for one real vector in the patch layout: Pacc = sum Re(c_n) u_n, Qacc = sum Im(c_n) u_n with
//   u_1 = v, u_2 = A' u_1 (shifted and scaled: A' = a A + b), u_{n+1} = 2 A' u_n - u_{n-1}
//   A = CB diag(Ebar)  (TRANSPOSED: diag(Ebar) CB^T),  e1 = a c^4 Ebar (the scale of A' and the c^4 of the factored colours ride on it)
// The arithmetic of kpm_series (kpm_sq_dev.h) with the registers cut to five vectors + one temporary: the doubled diagonal 2 e1 is made
// on the fly (an exact doubling: the same roundings), and the history term is built IN PLACE in the vector the new one overwrites.
template <int NS, bool TRANSPOSED, class APPLY>
__device__ __forceinline__ void series_lean(double (&Pacc)[NS], double (&Qacc)[NS], const double (&vin)[NS], const double (&e1)[NS],
                                            const double2 *c, int order, double b, APPLY &&apply) {
    double ua[NS], ub[NS];
    {
        const double2 c0 = c[0];
#pragma unroll
        for (int q = 0; q < NS; ++q) { Pacc[q] = c0.x * vin[q]; Qacc[q] = c0.y * vin[q]; ua[q] = vin[q]; ub[q] = 0.0; }
    }
    // one step: un = the latest vector, um = the one before (overwritten by the new one); tw = 1 for the first step (u_2 = A' u_1), 2 after
    auto step = [&](double (&un)[NS], double (&um)[NS], double tw, double bb, bool first, const double2 cprev, bool have_prev) {
        double w[NS];
#pragma unroll
        for (int q = 0; q < NS; ++q) {
            w[q] = TRANSPOSED ? un[q] : (tw * e1[q]) * un[q];
            um[q] = first ? bb * un[q] : bb * un[q] + um[q];              // the history term b' u_n + u_{n-1}  (b' = b or 2 b)
            if (have_prev) { Pacc[q] += cprev.x * un[q]; Qacc[q] += cprev.y * un[q]; }      // the sums of the vector made one step ago
        }
        apply(w);
#pragma unroll
        for (int q = 0; q < NS; ++q) um[q] = TRANSPOSED ? (tw * e1[q]) * w[q] - um[q] : w[q] - um[q];
    };
    const double b2 = 2.0 * b;
    if (order >= 2) step(ua, ub, 1.0, b, true, make_double2(0.0, 0.0), false);          // u_2 in ub (its coefficient: c[1])
    int n = 3;
    for (; n + 1 <= order; n += 2) {
        step(ub, ua, 2.0, b2, false, c[n - 2], true);          // u_n in ua; the sums of u_{n-1} (coefficient c[n-2]) ride along
        step(ua, ub, 2.0, b2, false, c[n - 1], true);          // u_{n+1} in ub
    }
    if (n <= order) {
        step(ub, ua, 2.0, b2, false, c[n - 2], true);          // u_order in ua
        const double2 cl = c[order - 1];
#pragma unroll
        for (int q = 0; q < NS; ++q) { Pacc[q] += cl.x * ua[q]; Qacc[q] += cl.y * ua[q]; }
    } else if (order >= 2) {
        const double2 cl = c[order - 1];
#pragma unroll
        for (int q = 0; q < NS; ++q) { Pacc[q] += cl.x * ub[q]; Qacc[q] += cl.y * ub[q]; }
    }
}

// rz_part != nullptr (round 6, the p/x-fused iteration of the patch-form lattices): the partial sums of r.(P^-1 r) in frequency space —
// Parseval for the twisted transform, a.b = (1/L) sum_k conj(a_k) b_k; a wave holds the real or the imaginary parts of one frequency, whose
// mirror L-1-k contributes the same — in slot 2 y + wave of this right-hand side (y = the block's position in the longest-first schedule);
// every block also clears the slots beyond 2 Lo2 that are its share.  A chain whose expansion is inactive hands over the r.r partials of the
// residual update instead (kernels of cg_fast_shared.inc: the same contract).
template <class LAT>
__global__ void __launch_bounds__(2 * LAT::NW * WAVE) k_kpm_cheb_pg(double2 *__restrict__ nu, KpmDev K, int N, int Ls, int Lo2, const CgState *state,
                                                          double *__restrict__ rz_part, int nrz, int Ltau, const double *__restrict__ rr_part,
                                                          const int *__restrict__ pgb) {
    constexpr int NS = LAT::NS, NW = LAT::NW, NT = NW * WAVE;      // NT threads hold the real parts, NT the imaginary parts (wv = 0 / 1)
    __shared__ double xch[2][NS * NT];
    __shared__ double xbuf[2][LAT::XB2_DOUBLES];                    // (several wavefronts per slice: the patch edges through LDS, pgrid_dev.h; two buffers per part: a series sweeps in ONE direction, a barrier separates the two series)
    __shared__ double red[2][NW];
    const int wv = threadIdx.x / NT, lane = threadIdx.x % NT;
    const int rhs = blockIdx.x;               // x = right-hand side, y = frequency in longest-first order
    if (state && state[2 * rhs].done) return;
    const KpmChainView V = kpm_chain_view(K, rhs, N);
    const int w = V.wsched[blockIdx.y];
    const int order = V.order[w];
    const double2 *c = K.coeff + V.coff[w];
    auto put_rz = [&](double dot_wave) {      // thread 0 of each part: this block's two slots, and its share of the slots beyond the schedule
        if (!rz_part || lane != 0) return;
        const int bid = 2 * (int)blockIdx.y + wv, nb = 2 * (int)gridDim.y;
        double *slots = rz_part + (size_t)rhs * nrz;
        slots[bid] = V.active ? dot_wave / (double)Ltau : (bid < Ltau ? rr_part[(size_t)rhs * Ltau + bid] : 0.0);
        for (int qq = nb + bid; qq < nrz; qq += nb) slots[qq] = 0.0;
    };
    if (order == 1) {
        // A series of order 1 is its leading coefficient: z_w = c0 (conj(c0) r_w) — no checkerboard, no patch layout.  That is most
        // frequencies (order_w ~ 1 / phi_w: 45 of 80 at L_tau = 160), and a block of its own for each would hold a wave slot for a few
        // microseconds of launch and dependent loads to do a microsecond of arithmetic: the FIRST ORD1_BLOCKS order-1 positions of the
        // (longest-first) schedule share all of them, the rest leave at once.  (products first, then the sums: the roundings of the
        // general path)
        constexpr int ORD1_BLOCKS = 4;
        const int y = (int)blockIdx.y;
        if (y >= ORD1_BLOCKS && V.order[V.wsched[y - ORD1_BLOCKS]] == 1) { put_rz(0.0); return; }
        double dot = 0.0;
        for (int yy = y; yy < Lo2; yy += ORD1_BLOCKS) {
            const int ww = V.wsched[yy];
            const double2 c0 = K.coeff[V.coff[ww]];
            const double wgt = ((Ltau & 1) && ww == Lo2 - 1) ? 1.0 : 2.0;
            double2 *uc = nu + ((size_t)rhs * Lo2 + ww) * N;
            double d1 = 0.0;
            for (int i = threadIdx.x; i < N; i += 2 * NT) {
                const double2 v = uc[i];
                const double mr = __dadd_rn(__dmul_rn(c0.x, v.x), __dmul_rn(c0.y, v.y)), mi = __dsub_rn(__dmul_rn(c0.x, v.y), __dmul_rn(c0.y, v.x));
                const double zr = __dsub_rn(__dmul_rn(c0.x, mr), __dmul_rn(c0.y, mi)), zi = __dadd_rn(__dmul_rn(c0.x, mi), __dmul_rn(c0.y, mr));
                uc[i] = make_double2(zr, zi);
                d1 += v.x * zr + v.y * zi;
            }
            dot += wgt * d1;
        }
        if (rz_part) put_rz(pg_group_sum<NW>(dot, red[wv], lane));      // (rz_part is uniform over the block)
        return;
    }
    double *u = reinterpret_cast<double *>(nu + ((size_t)rhs * Lo2 + w) * N);
    const bool act = lane < LAT::lanes(Ls);
    pgrid::Ctx X = LAT::make_ctx(lane, Ls, V.cbar[0], V.sbar[0], xbuf[wv], 1);
    pg_fill_tab<LAT>(X, lane, Ls, N, pgb, V.cbar, V.sbar);
    const double a = V.a * X.ks, b = V.b;
    int site[NS];
    double e1[NS], vin[NS], Pa[NS], Qa[NS];
#pragma unroll
    for (int q = 0; q < NS; ++q) {
        site[q] = LAT::site_of(lane, q, Ls);
        e1[q] = a * V.Ebar[site[q]];
        vin[q] = u[2 * site[q] + wv];
    }
    // M^-T[w,w]: conjugated coefficients, transposed A   (KPMPreconditioners.jl:621-648)
    series_lean<NS, true>(Pa, Qa, vin, e1, c, order, b, [&X](double (&v)[NS]) { LAT::template apply<true>(v, X); });
#pragma unroll
    for (int q = 0; q < NS; ++q) xch[wv][q * NT + lane] = Qa[q];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NS; ++q) {
        const double Qo = xch[wv ^ 1][q * NT + lane];
        vin[q] = (wv == 0) ? Pa[q] + Qo : Pa[q] - Qo;          // (vin now holds the middle vector: the input of the second series)
    }
    __syncthreads();
    // M^-1[w,w]   (:650-677)
    series_lean<NS, false>(Pa, Qa, vin, e1, c, order, b, [&X](double (&v)[NS]) { LAT::template apply<false>(v, X); });
#pragma unroll
    for (int q = 0; q < NS; ++q) xch[wv][q * NT + lane] = Qa[q];
    __syncthreads();
    double dot = 0.0;
#pragma unroll
    for (int q = 0; q < NS; ++q) {
        const double Qo = xch[wv ^ 1][q * NT + lane];
        const double res = (wv == 0) ? Pa[q] - Qo : Pa[q] + Qo;
        if (act) {
            // (the input r_w is read back from memory — it is overwritten only now; keeping it in registers through both series costs NS of them)
            if (rz_part) dot += u[2 * site[q] + wv] * res;
            u[2 * site[q] + wv] = res;
        }
    }
    if (rz_part) put_rz((((Ltau & 1) && w == Lo2 - 1) ? 1.0 : 2.0) * pg_group_sum<NW>(dot, red[wv], lane));
}
// every wave reduces ALL partials itself: same loads, same tree => the same bits in every wave (kernels.hip: reduce_partials)
__device__ __forceinline__ double pg_reduce_partials(const double *p, int n, int lane) {
    double a = 0.0;
    for (int i = lane; i < n; i += WAVE) a += p[i];
    return pg_wave_sum(a);
}

// The first kernel of a CG iteration (IterativeSolvers.jl:153-234 / :236-315; the contract of kernels.hip: k_cg_ap — stop test of the
// previous iteration, beta, p = (z | r) + beta p, z = M^T (M p), the partial sums of p.z, the other copy of the state) for a large
// square lattice: ONE wavefront takes Tmax consecutive time slices of one right-hand side and walks through them with p(t-1), w(t-1)
// in registers — every slice of z|r, p_old and exp(-dtau V) is read once (+ two halo slices per chunk), p_new and z are written once;
// the two checkerboard sweeps per slice (M: forward on E(t) p(t-1); M^T: reverse on w(t), then E(t)) are patch sweeps in registers.
// The generic kernel it replaces takes one slice per workgroup: three slices of z|r and p_old read per slice written, and
// every colour of its sweeps is an LDS round trip between two barriers (283 us per iteration of 72 right-hand sides at L = 32, where
// the bytes that must move take ~100).
//   w(t)   = p(t) - sg(t) c^4 S(E(t) p(t-1))            sg(t) = -1 at t = 0 (antiperiodic), S / S^T: the sweep without its c^4
//   z(t-1) = w(t-1) - sg(t) c^4 E(t) S^T(w(t))
// PX (round 6): the p/x-fused preconditioned iteration — the search direction arrives READY in B.p (slot 0: the inverse tau-transform of the
// previous iteration formed p = P^-1 r + beta p and applied x += alpha p in its epilogue, dft_mfma.hip: PxFuse): this kernel reads p (own slices
// + two halo slices) and exp(-dtau V), writes z and the p.z partials and keeps the scalar state machine; no P^-1 r, no p_old, no p_new.
template <class LAT, bool PX>
__global__ void __launch_bounds__(LAT::NW * WAVE) k_cg_ap_pg(CgBufs B, ModelDev m, int parity, int Ls, int Tmax, const int *__restrict__ pgb) {
    constexpr int NS = LAT::NS, NW = LAT::NW;
    __shared__ double xbuf[LAT::XB_DOUBLES];
    __shared__ double red[NW];
    const int N = m.N, L = m.L, lane = threadIdx.x;      // (several wavefronts per slice: the thread's index in the slice)
    const int nch = (L + Tmax - 1) / Tmax;
    const int rhs = blockIdx.x / nch, ch = blockIdx.x - rhs * nch;
    const int t0 = ch * Tmax, T = (L - t0 < Tmax) ? L - t0 : Tmax;
    const size_t ndim = (size_t)N * L;
    CgState *st2 = B.state + 2 * rhs;
    const CgState S = st2[parity];
    CgState *Sout = st2 + (parity ^ 1);
    if (S.done) {
        if (ch == 0 && lane == 0) *Sout = S;
        return;
    }
    const CgParams P = B.params;
    const long long seq = S.seq;
    const bool first = (seq == 0);
    double beta = 0.0, rho = S.rho, kmin = S.kmin, eps = S.eps;
    if (!first) {
        // stop test of iteration `seq` (IterativeSolvers.jl:286-295 / :211-219)
        const double rr = pg_reduce_partials(B.rr + (size_t)rhs * L, L, lane & (WAVE - 1));
        eps = sqrt(rr) / S.normb;
        const double qq = 2.0 * (double)seq / log(2.0 * S.eps0 / eps);
        const double val = qq * qq;
        kmin = (val > kmin) ? val : kmin;
        int done = 0;
        if (eps < P.tol) done = 1;
        else if (kmin > P.kmax) done = 2;
        else if (seq >= P.maxiter) done = 3;
        if (ch == 0 && lane == 0 && P.record_hist) B.hist[(size_t)rhs * P.hist_stride + seq] = eps;
        if (done) {
            if (ch == 0 && lane == 0) {
                CgState o = S;
                o.kmin = kmin; o.eps = eps; o.seq = seq + 1; o.iters = seq; o.done = done;
                *Sout = o;
            }
            return;
        }
        const double rho_new = P.use_prec ? pg_reduce_partials(B.rz + (size_t)rhs * B.nrz, B.nrz, lane & (WAVE - 1)) : rr;
        beta = rho_new / S.rho;            // :222-223 / :303-304
        rho = rho_new;
    }
    const double *src = (P.use_prec ? B.zp : B.r) + (size_t)rhs * ndim;
    const double *pold = B.p + ((size_t)(PX ? 0 : parity) * B.nrhs + rhs) * ndim;
    double *pnew = B.p + ((size_t)(parity ^ 1) * B.nrhs + rhs) * ndim;
    double *z = B.z + (size_t)rhs * ndim;
    const double *Ech = m.E + (size_t)(rhs % m.nchains) * m.E_chain_stride;
    const bool act = lane < LAT::lanes(Ls);
    pgrid::Ctx X = LAT::make_ctx(lane, Ls, m.c_uni, m.s_uni, xbuf);
    pg_fill_tab<LAT>(X, lane, Ls, N, pgb, m.c, m.s);
    int site[NS];
    bool dot[NS];
#pragma unroll
    for (int q = 0; q < NS; ++q) { site[q] = LAT::site_of(lane, q, Ls); dot[q] = act && site[q] >= B.dot_lo && site[q] < B.dot_hi; }
    auto wrap = [L](int t) { return (t < 0) ? t + L : ((t >= L) ? t - L : t); };
    // p(t) = (z | r)(t) + beta p_old(t)   (:229-230 / :309-310); the first iteration: p0 as the init kernel stored it
    auto load_raw = [&](int t, double (&sv)[NS], double (&qv)[NS]) {
#pragma unroll
        for (int q = 0; q < NS; ++q) {
            const size_t i = (size_t)t * N + site[q];
            qv[q] = pold[i];
            sv[q] = (PX || first) ? 0.0 : src[i];
        }
    };
    auto load_e = [&](int t, double (&ev)[NS]) {
        const double *Et = Ech + (size_t)t * m.E_tau_stride;
#pragma unroll
        for (int q = 0; q < NS; ++q) ev[q] = Et[site[q]];
    };
    double pprev[NS], wprev[NS];
    {
        double s0[NS], q0[NS];
        load_raw(wrap(t0 - 1), s0, q0);
#pragma unroll
        for (int q = 0; q < NS; ++q) pprev[q] = (PX || first) ? q0[q] : s0[q] + beta * q0[q];
    }
    // the loads of slice t + 1 are issued before the sweeps of slice t: a wave is alone or nearly alone on its SIMD (a slice is 2 x NS
    // registers per vector), nothing else would hide the round trip
    double Sn[NS], Qn[NS], En[NS];
    load_raw(t0, Sn, Qn);
    load_e(t0, En);
    double acc = 0.0;
#pragma unroll 1
    for (int j = 0; j <= T; ++j) {
        const int t = wrap(t0 + j);
        const double sg = (t == 0) ? -X.ks : X.ks;         // the sign of the antiperiodic boundary with the c^4 (c^3) of the factored colours
        double pcur[NS], wcur[NS], Ecur[NS];
#pragma unroll
        for (int q = 0; q < NS; ++q) {
            pcur[q] = (PX || first) ? Qn[q] : Sn[q] + beta * Qn[q];
            Ecur[q] = En[q];
            wcur[q] = Ecur[q] * pprev[q];
        }
        if (j < T) {
            const int tn = wrap(t0 + j + 1);
            load_raw(tn, Sn, Qn);
            load_e(tn, En);
            if (act && !PX) {
#pragma unroll
                for (int q = 0; q < NS; ++q) pnew[(size_t)t * N + site[q]] = pcur[q];
            }
        }
        LAT::template apply<false>(wcur, X);
#pragma unroll
        for (int q = 0; q < NS; ++q) wcur[q] = pcur[q] - sg * wcur[q];
        if (j > 0) {
            double g[NS];
#pragma unroll
            for (int q = 0; q < NS; ++q) g[q] = wcur[q];
            LAT::template apply<true>(g, X);
            const int tz = wrap(t0 + j - 1);
#pragma unroll
            for (int q = 0; q < NS; ++q) {
                const double zz = wprev[q] - sg * Ecur[q] * g[q];
                if (act) z[(size_t)tz * N + site[q]] = zz;
                if (dot[q]) acc += pprev[q] * zz;
            }
        }
#pragma unroll
        for (int q = 0; q < NS; ++q) { pprev[q] = pcur[q]; wprev[q] = wcur[q]; }
    }
    acc = pg_group_sum<NW>(acc, red, lane);
    if (lane == 0) {
        // one partial sum per chunk, in the slot of its first slice; the slots of its other slices are zero (npap = L: the generic family's layout)
        double *pap = B.pap + (size_t)rhs * B.npap;
        pap[t0] = acc;
        for (int j = 1; j < T; ++j) pap[t0 + j] = 0.0;
        if (ch == 0) {
            CgState o = S;
            o.rho = rho; o.kmin = kmin; o.eps = eps; o.seq = seq + 1; o.iters = seq; o.done = 0;
            *Sout = o;
        }
    }
}

// y = M v (WHICH 0), M^T v (1), M^T M v (2) — mulM!, mulMT!, mulMTM! (HolsteinModels.jl:569-684) in the patch layout, one wavefront per
// chunk of Tmax slices of a vector; the arithmetic of the generic k_mul (kernels.hip) with register sweeps:
//   (M v)(t)   = v(t) - sg(t) c^k S(E(t) v(t-1))          (M^T v)(t) = v(t) - sg(t+1) c^k E(t+1) S^T(v(t+1))
template <class LAT, int WHICH>
__global__ void __launch_bounds__(LAT::NW * WAVE) k_mul_pg(double *__restrict__ y, const double *__restrict__ v, ModelDev m, int Ls, int Tmax,
                                                           const int *__restrict__ pgb) {
    constexpr int NS = LAT::NS;
    __shared__ double xbuf[LAT::XB_DOUBLES];
    const int N = m.N, L = m.L, lane = threadIdx.x;
    const int nch = (L + Tmax - 1) / Tmax;
    const int vecno = blockIdx.x / nch, ch = blockIdx.x - vecno * nch;
    const int t0 = ch * Tmax, T = (L - t0 < Tmax) ? L - t0 : Tmax;
    const size_t vec = (size_t)vecno * (size_t)N * (size_t)L;
    const double *vv = v + vec;
    double *yy = y + vec;
    const double *Ech = m.E + (size_t)(vecno % m.nchains) * m.E_chain_stride;
    const bool act = lane < LAT::lanes(Ls);
    pgrid::Ctx X = LAT::make_ctx(lane, Ls, m.c_uni, m.s_uni, xbuf);
    pg_fill_tab<LAT>(X, lane, Ls, N, pgb, m.c, m.s);
    int site[NS];
#pragma unroll
    for (int q = 0; q < NS; ++q) site[q] = LAT::site_of(lane, q, Ls);
    auto wrap = [L](int t) { return (t < 0) ? t + L : ((t >= L) ? t - L : t); };
    auto load = [&](const double *base, int t, double (&a)[NS]) {
#pragma unroll
        for (int q = 0; q < NS; ++q) a[q] = base[(size_t)t * N + site[q]];
    };
    auto sgn = [&X](int t) { return (t == 0) ? -X.ks : X.ks; };
    if constexpr (WHICH == 0) {
        double vprev[NS];
        load(vv, wrap(t0 - 1), vprev);
#pragma unroll 1
        for (int j = 0; j < T; ++j) {
            const int t = t0 + j;
            double vc[NS], f[NS];
            load(vv, t, vc);
            const double *Et = Ech + (size_t)t * m.E_tau_stride;
#pragma unroll
            for (int q = 0; q < NS; ++q) f[q] = Et[site[q]] * vprev[q];
            LAT::template apply<false>(f, X);
            const double sg = sgn(t);
#pragma unroll
            for (int q = 0; q < NS; ++q) { if (act) yy[(size_t)t * N + site[q]] = vc[q] - sg * f[q]; vprev[q] = vc[q]; }
        }
    } else if constexpr (WHICH == 1) {
        double vc[NS];
        load(vv, t0, vc);
#pragma unroll 1
        for (int j = 0; j < T; ++j) {
            const int t = t0 + j, tp = wrap(t + 1);
            double vn[NS], g[NS];
            load(vv, tp, vn);
            const double *Ep = Ech + (size_t)tp * m.E_tau_stride;
#pragma unroll
            for (int q = 0; q < NS; ++q) g[q] = vn[q];
            LAT::template apply<true>(g, X);
            const double sg = sgn(tp);
#pragma unroll
            for (int q = 0; q < NS; ++q) { if (act) yy[(size_t)t * N + site[q]] = vc[q] - sg * Ep[site[q]] * g[q]; vc[q] = vn[q]; }
        }
    } else {
        // w(t) = v(t) - sg(t) c^k S(E(t) v(t-1)) for t0 .. t0+T; y(t-1) = w(t-1) - sg(t) c^k E(t) S^T(w(t))
        double vprev[NS], wprev[NS];
        load(vv, wrap(t0 - 1), vprev);
#pragma unroll 1
        for (int j = 0; j <= T; ++j) {
            const int t = wrap(t0 + j);
            double vc[NS], wc[NS], Ec[NS];
            load(vv, t, vc);
            const double *Et = Ech + (size_t)t * m.E_tau_stride;
#pragma unroll
            for (int q = 0; q < NS; ++q) { Ec[q] = Et[site[q]]; wc[q] = Ec[q] * vprev[q]; }
            LAT::template apply<false>(wc, X);
            const double sg = sgn(t);
#pragma unroll
            for (int q = 0; q < NS; ++q) wc[q] = vc[q] - sg * wc[q];
            if (j > 0) {
                double g[NS];
#pragma unroll
                for (int q = 0; q < NS; ++q) g[q] = wc[q];
                LAT::template apply<true>(g, X);
                const int tz = wrap(t0 + j - 1);
#pragma unroll
                for (int q = 0; q < NS; ++q) if (act) yy[(size_t)tz * N + site[q]] = wprev[q] - sg * Ec[q] * g[q];
            }
#pragma unroll
            for (int q = 0; q < NS; ++q) { vprev[q] = vc[q]; wprev[q] = wc[q]; }
        }
    }
}

int pg_check(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { elph_set_error("launch %s failed: %s", what, hipGetErrorString(e)); return ELPH_E_HIP; }
    return ELPH_OK;
}

}   // namespace

// The patch shape of ONE launch.  The memory layout of a vector does not depend on it (a shape only says which thread holds which sites), so a launch may
// take another shape than the handle's: 4 x 4 patches on one wavefront (28 x 28, 32 x 32) have the fewest instructions per site and the shortest
// dependent chain — one right-hand side: 89 us per preconditioned iteration against 96 as 2 x 2 patches on four wavefronts — but hold ONE wave per
// SIMD (16 registers per vector), and a large batch is then latency-bound.  Measured (profiles/r06/patch_shape_by_batch_size.log, 32 x 32): the
// recursion 79 -> 71 us at 48 right-hand sides, 96 -> 76 at 64, 133 -> 113 at 96, 176 -> 151 at 128 (24, 32: 63 -> 68, 66 -> 71: the other way); the
// p/x-fused k_cg_ap_pg 70 -> 61, 100 -> 92, 157 -> 139; the iteration 250 -> 233, 318 -> 291, 477 -> 443, 589 -> 546; 28 x 28 at 64: 281 -> 261.  The
// UNFUSED k_cg_ap_pg and k_mul_pg keep the handle's shape (72 right-hand sides: 184 -> 203 us the wrong way, 96: 273 -> 252).
// `big`: a launch that may switch (the recursion, the fused k_cg_ap).  ELPH_PG_2X2_FROM=n: from n right-hand sides [48] (A/B; 0: never).
static void pg_launch_shape(const elph_handle_s *h, int nrhs, bool big, int *px, int *py, int *nw) {
    *px = h->pg_PX; *py = h->pg_PY; *nw = h->pg_NW > 1 ? h->pg_NW : 1;
    if (!big || *nw != 1) return;
    const char *e = getenv("ELPH_PG_2X2_FROM"), *em = getenv("ELPH_PG_MW");
    const int from = e ? atoi(e) : 48;
    if (from <= 0 || nrhs < from || (em && em[0] == '0')) return;
    if (h->pg_kind == 1 && *px * *py >= 16 && ((h->pg_L / 2) * (h->pg_L / 2) + 63) / 64 <= 6 && h->pg_uniform_c) {      // 4 x 4 (28, 32), 2 x 10 (30: 96 right-hand sides on two streams 477 -> 415 us), 4 x 6 (36: 472 -> 459 at 64)
        *px = 2; *py = 2; *nw = ((h->pg_L / 2) * (h->pg_L / 2) + 63) / 64;      // 28: 196 threads, 32: 256 — four wavefronts; 30: 225 — four; 36: 324 — six
    } else if (h->pg_kind == 2 && *px == 4 && *py == 2) {
        // honeycomb 20 x 20 cells (4 x 2 cells per lane: the other 16-register shape) as 2 x 2 cells on two wavefronts: 96 right-hand sides 315 -> 292 us per
        // iteration, on two streams 295 -> 247; 18 x 18 (3 x 2 cells) is indifferent, 24 x 24 (3 x 3) LOSES as 2 x 2 cells on three (538 -> 590 at 128) and
        // keeps its shape (profiles/r06/patch_shape_by_batch_size.log)
        *px = 2; *py = 2; *nw = ((h->pg_L / 2) * (h->pg_L / 2) + 63) / 64;
    }
}

// Can the per-frequency recursion of this handle run in the patch layout?  (ELPH_NO_PG=1: the generic kernel, the A/B — read per call)
bool elph_pg_cheb_usable(const elph_handle_s *h) {
    const char *e = getenv("ELPH_NO_PG");
    return h->pg_L > 0 && (h->pg_uniform || elph_pg_disorder_ok(h)) && h->kind == ELPH_MODEL_HOLSTEIN && !(e && e[0] == '1');
}

// hopping disorder on this handle's patch shape (square lattices whose (cosh, sinh) table fits the LDS: pgrid::patch_takes_disorder; ELPH_PG_DIS=0:
// the generic kernels — the A/B, read per call)
bool elph_pg_disorder_ok(const elph_handle_s *h) {
    const char *e = getenv("ELPH_PG_DIS");
    // (measured, profiles/r06/hopping_disorder_patch_kernels_with_tables.log: 18 x 18 — lane-program mat-vec, only the recursion would move — loses to
    //  the Re / Im recursion through the LDS slab, 214 -> 252 us at 96 right-hand sides; 20 x 20 and 22 x 22, the same family, gain: 241 -> 195, 313 -> 266)
    if (h->fast && h->pg_L == 18) return false;
    return h->pg_L > 0 && h->pg_kind == 1 && h->d_pg_bond && pgrid::patch_takes_disorder(h->pg_PX, h->pg_PY, h->pg_NW > 1 ? h->pg_NW : 1) && !(e && e[0] == '0');
}

// nu (d_nu: [nrhs][Lo2][N] complex) <- P^-1 in frequency space, every (right-hand side, frequency) a block of two wavefronts
int elph_pg_kpm_cheb(elph_handle_s *h, int nrhs, const CgState *st, double *rz_part, int nrz, const double *rr_part) {
    KpmDev K = elph_kpm_dev(h);
    const int Lo2 = (int)((h->L + 1) / 2), N = (int)h->N, Ls = h->pg_L;
    if (rz_part && 2 * Lo2 > nrz) { elph_set_error("k_kpm_cheb_pg: %d r.z slots needed, %d available", 2 * Lo2, nrz); return ELPH_E_STATE; }
    const dim3 grid((unsigned)nrhs, (unsigned)Lo2);
#define PG_CHEB(LAT) hipLaunchKernelGGL((k_kpm_cheb_pg<LAT>), grid, dim3(2 * LAT::NW * WAVE), LAT::TAB_BYTES, h->stream, h->d_nu, K, N, Ls, Lo2, st, rz_part, nrz, (int)h->L, rr_part, h->d_pg_bond)
    int px, py, nw;
    pg_launch_shape(h, nrhs, true, &px, &py, &nw);
    if (!h->pg_uniform) {        // hopping disorder: the table variants (elph_pg_cheb_usable has checked the shape)
        if (!elph_pg_disorder_ok(h)) { elph_set_error("k_kpm_cheb_pg: hopping disorder on a patch shape without a table variant"); return ELPH_E_UNSUPPORTED; }
        if (nw == 2) PG_CHEB(M22_2D);
        else if (nw == 3) PG_CHEB(M22_3D);
        else if (nw == 4) PG_CHEB(M22_4D);
        else if (nw == 5) PG_CHEB(M22_5D);
        else if (px == 4) PG_CHEB(SQ44D);
        else if (py == 6) PG_CHEB(SQ26D);
        else PG_CHEB(SQ24D);
    }
    else if (h->pg_kind == 1 && nw > 1) {
        if (px == 2 && py == 2 && nw == 2) PG_CHEB(M22_2);
        else if (px == 2 && py == 2 && nw == 3) PG_CHEB(M22_3);
        else if (px == 2 && py == 2 && nw == 4) PG_CHEB(M22_4);
        else if (px == 2 && py == 2 && nw == 5) PG_CHEB(M22_5);
        else if (px == 2 && py == 2 && nw == 6) PG_CHEB(M22_6);
        else if (px == 4 && py == 4 && nw == 2) PG_CHEB(M44_2);
        else if (px == 4 && py == 4 && nw == 3) PG_CHEB(M44_3);
        else if (px == 4 && py == 4 && nw == 4) PG_CHEB(M44_4);
        else { elph_set_error("patch kernels: no instantiation for %d x %d patches on %d wavefronts", px, py, nw); return ELPH_E_UNSUPPORTED; }
    }
    else if (h->pg_kind == 2 && nw > 1) {
        if (px == 3 && nw == 2) PG_CHEB(H33_2);
        else if (px == 3 && nw == 3) PG_CHEB(H33_3);
        else if (px == 3 && nw == 4) PG_CHEB(H33_4);
        else if (px == 2 && nw == 2) PG_CHEB(H22_2);
        else if (px == 2 && nw == 3) PG_CHEB(H22_3);
        else if (px == 2 && nw == 4) PG_CHEB(H22_4);
        else { elph_set_error("patch kernels: no honeycomb instantiation for %d x %d cells on %d wavefronts", px, py, nw); return ELPH_E_UNSUPPORTED; }
    }
    else if (h->pg_kind == 1 && px == 4 && py == 4) PG_CHEB(SQ44);
    else if (h->pg_kind == 1 && px == 2 && py == 6) PG_CHEB(SQ26);
    else if (h->pg_kind == 1 && px == 2 && py == 4) PG_CHEB(SQ24);
    else if (h->pg_kind == 1 && px == 2 && py == 10) PG_CHEB(SQ2A);
    else if (h->pg_kind == 1 && px == 4 && py == 6) PG_CHEB(SQ46);
    else if (h->pg_kind == 2 && px == 3 && py == 2) PG_CHEB(HC32);
    else if (h->pg_kind == 2 && px == 4 && py == 2) PG_CHEB(HC42);
    else if (h->pg_kind == 2 && px == 3 && py == 3) PG_CHEB(HC33);
    else if (h->pg_kind == 3 && px == 2 && py == 2) PG_CHEB(TR22);
    else if (h->pg_kind == 3 && px == 2 && py == 4) PG_CHEB(TR24);
    else if (h->pg_kind == 3 && px == 2 && py == 6) PG_CHEB(TR26);
    else if (h->pg_kind == 3 && px == 4 && py == 4) PG_CHEB(TR44);
    else { elph_set_error("k_kpm_cheb_pg: no instantiation for kind %d, %d x %d patches", h->pg_kind, px, py); return ELPH_E_UNSUPPORTED; }
#undef PG_CHEB
    return pg_check("k_kpm_cheb_pg");
}

// the mat-vec kernel of the CG iteration in the patch layout (generic family only: the lane-program family has its own chunked kernel)
bool elph_pg_ap_usable(const elph_handle_s *h) {
    const char *e = getenv("ELPH_NO_PG");
    return h->pg_L > 0 && !h->fast && h->kind == ELPH_MODEL_HOLSTEIN && !(e && e[0] == '1');
}

int elph_pg_cg_ap(elph_handle_s *h, const CgBufs &B, const ModelDev &m, int nrhs, int parity, bool fused) {
    if (!m.uniform && !elph_pg_disorder_ok(h)) return ELPH_E_UNSUPPORTED;
    if (B.npap != (int)h->L) { elph_set_error("k_cg_ap_pg: one p.z slot per time slice expected"); return ELPH_E_UNSUPPORTED; }
    // slices per wave: the SHORTEST chunk whose waves still fit the chip in one round — 1024 SIMDs x the waves a SIMD holds of this
    // instantiation (one for 12 or 16 sites per lane: 256 + ~100 registers with the prefetched slice; two for 8) — a second, partly
    // filled round costs more than the two halo slices per chunk (measured at L = 32, 72 right-hand sides: 16 slices per wave = 720
    // waves 98 us, 10 = 1152 waves 140 us; profiles/r04/pgrid_large_lattices.log); beyond one round of 40-slice chunks: 20
    const int L = (int)h->L;
    static const int forceT = []() { const char *e = getenv("ELPH_PG_T"); return e ? atoi(e) : 0; }();
    int px, py, nw;
    pg_launch_shape(h, nrhs, fused, &px, &py, &nw);
    const long long slots = 1024LL * ((h->pg_kind != 2 && px * py <= 8) ? 2 : 1) / nw;      // (a slice of several wavefronts holds as many slots)
    int T = 20;
    for (int c : {1, 2, 4, 5, 8, 10, 16, 20, 32, 40}) { if ((long long)nrhs * ((L + c - 1) / c) <= slots) { T = c; break; } }
    if (forceT > 0) T = forceT;
    T = std::max(1, std::min(T, L));
    const int nch = (L + T - 1) / T;
    const dim3 grid((unsigned)(nrhs * nch));
    const int Ls = h->pg_L;
#define PG_AP(LAT)                                                                                                        \
    do {                                                                                                                  \
        if (fused) hipLaunchKernelGGL((k_cg_ap_pg<LAT, true>), grid, dim3(LAT::NW * WAVE), LAT::TAB_BYTES, h->stream, B, m, parity, Ls, T, h->d_pg_bond);              \
        else hipLaunchKernelGGL((k_cg_ap_pg<LAT, false>), grid, dim3(LAT::NW * WAVE), LAT::TAB_BYTES, h->stream, B, m, parity, Ls, T, h->d_pg_bond);                \
    } while (0)
    if (!m.uniform) {
        if (nw == 2) PG_AP(M22_2D);
        else if (nw == 3) PG_AP(M22_3D);
        else if (nw == 4) PG_AP(M22_4D);
        else if (nw == 5) PG_AP(M22_5D);
        else if (px == 4) PG_AP(SQ44D);
        else if (py == 6) PG_AP(SQ26D);
        else PG_AP(SQ24D);
    }
    else if (h->pg_kind == 1 && nw > 1) {
        if (px == 2 && py == 2 && nw == 2) PG_AP(M22_2);
        else if (px == 2 && py == 2 && nw == 3) PG_AP(M22_3);
        else if (px == 2 && py == 2 && nw == 4) PG_AP(M22_4);
        else if (px == 2 && py == 2 && nw == 5) PG_AP(M22_5);
        else if (px == 2 && py == 2 && nw == 6) PG_AP(M22_6);
        else if (px == 4 && py == 4 && nw == 2) PG_AP(M44_2);
        else if (px == 4 && py == 4 && nw == 3) PG_AP(M44_3);
        else if (px == 4 && py == 4 && nw == 4) PG_AP(M44_4);
        else { elph_set_error("patch kernels: no instantiation for %d x %d patches on %d wavefronts", px, py, nw); return ELPH_E_UNSUPPORTED; }
    }
    else if (h->pg_kind == 2 && nw > 1) {
        if (px == 3 && nw == 2) PG_AP(H33_2);
        else if (px == 3 && nw == 3) PG_AP(H33_3);
        else if (px == 3 && nw == 4) PG_AP(H33_4);
        else if (px == 2 && nw == 2) PG_AP(H22_2);
        else if (px == 2 && nw == 3) PG_AP(H22_3);
        else if (px == 2 && nw == 4) PG_AP(H22_4);
        else { elph_set_error("patch kernels: no honeycomb instantiation for %d x %d cells on %d wavefronts", px, py, nw); return ELPH_E_UNSUPPORTED; }
    }
    else if (h->pg_kind == 1 && px == 4 && py == 4) PG_AP(SQ44);
    else if (h->pg_kind == 1 && px == 2 && py == 6) PG_AP(SQ26);
    else if (h->pg_kind == 1 && px == 2 && py == 4) PG_AP(SQ24);
    else if (h->pg_kind == 1 && px == 2 && py == 10) PG_AP(SQ2A);
    else if (h->pg_kind == 1 && px == 4 && py == 6) PG_AP(SQ46);
    else if (h->pg_kind == 2 && px == 3 && py == 2) PG_AP(HC32);
    else if (h->pg_kind == 2 && px == 4 && py == 2) PG_AP(HC42);
    else if (h->pg_kind == 2 && px == 3 && py == 3) PG_AP(HC33);
    else if (h->pg_kind == 3 && px == 2 && py == 2) PG_AP(TR22);
    else if (h->pg_kind == 3 && px == 2 && py == 4) PG_AP(TR24);
    else if (h->pg_kind == 3 && px == 2 && py == 6) PG_AP(TR26);
    else if (h->pg_kind == 3 && px == 4 && py == 4) PG_AP(TR44);
    else { elph_set_error("k_cg_ap_pg: no instantiation for kind %d, %d x %d patches", h->pg_kind, px, py); return ELPH_E_UNSUPPORTED; }
#undef PG_AP
    return pg_check("k_cg_ap_pg");
}

// mulM!, mulMT!, mulMTM! in the patch layout (generic family only)
bool elph_pg_mul_usable(const elph_handle_s *h) { return elph_pg_ap_usable(h); }

int elph_pg_mul(elph_handle_s *h, const ModelDev &m, int which, double *yS, const double *vS, int nvec) {
    if (!m.uniform && !elph_pg_disorder_ok(h)) return ELPH_E_UNSUPPORTED;
    const int L = (int)h->L, Ls = h->pg_L;
    int T = 20;
    for (int c : {1, 2, 4, 5, 8, 10, 16, 20}) { if ((long long)nvec * ((L + c - 1) / c) <= 2048) { T = c; break; } }
    T = std::max(1, std::min(T, L));
    const dim3 grid((unsigned)(nvec * ((L + T - 1) / T)));
#define PG_MUL(LAT)                                                                                                       \
    do {                                                                                                                  \
        if (which == 0) hipLaunchKernelGGL((k_mul_pg<LAT, 0>), grid, dim3(LAT::NW * WAVE), LAT::TAB_BYTES, h->stream, yS, vS, m, Ls, T, h->d_pg_bond);              \
        else if (which == 1) hipLaunchKernelGGL((k_mul_pg<LAT, 1>), grid, dim3(LAT::NW * WAVE), LAT::TAB_BYTES, h->stream, yS, vS, m, Ls, T, h->d_pg_bond);         \
        else hipLaunchKernelGGL((k_mul_pg<LAT, 2>), grid, dim3(LAT::NW * WAVE), LAT::TAB_BYTES, h->stream, yS, vS, m, Ls, T, h->d_pg_bond);                         \
    } while (0)
    int px, py, nw;
    pg_launch_shape(h, nvec, false, &px, &py, &nw);
    if (!m.uniform) {
        if (nw == 2) PG_MUL(M22_2D);
        else if (nw == 3) PG_MUL(M22_3D);
        else if (nw == 4) PG_MUL(M22_4D);
        else if (nw == 5) PG_MUL(M22_5D);
        else if (px == 4) PG_MUL(SQ44D);
        else if (py == 6) PG_MUL(SQ26D);
        else PG_MUL(SQ24D);
    }
    else if (h->pg_kind == 1 && nw > 1) {
        if (px == 2 && py == 2 && nw == 2) PG_MUL(M22_2);
        else if (px == 2 && py == 2 && nw == 3) PG_MUL(M22_3);
        else if (px == 2 && py == 2 && nw == 4) PG_MUL(M22_4);
        else if (px == 2 && py == 2 && nw == 5) PG_MUL(M22_5);
        else if (px == 2 && py == 2 && nw == 6) PG_MUL(M22_6);
        else if (px == 4 && py == 4 && nw == 2) PG_MUL(M44_2);
        else if (px == 4 && py == 4 && nw == 3) PG_MUL(M44_3);
        else if (px == 4 && py == 4 && nw == 4) PG_MUL(M44_4);
        else { elph_set_error("patch kernels: no instantiation for %d x %d patches on %d wavefronts", px, py, nw); return ELPH_E_UNSUPPORTED; }
    }
    else if (h->pg_kind == 2 && nw > 1) {
        if (px == 3 && nw == 2) PG_MUL(H33_2);
        else if (px == 3 && nw == 3) PG_MUL(H33_3);
        else if (px == 3 && nw == 4) PG_MUL(H33_4);
        else if (px == 2 && nw == 2) PG_MUL(H22_2);
        else if (px == 2 && nw == 3) PG_MUL(H22_3);
        else if (px == 2 && nw == 4) PG_MUL(H22_4);
        else { elph_set_error("patch kernels: no honeycomb instantiation for %d x %d cells on %d wavefronts", px, py, nw); return ELPH_E_UNSUPPORTED; }
    }
    else if (h->pg_kind == 1 && px == 4 && py == 4) PG_MUL(SQ44);
    else if (h->pg_kind == 1 && px == 2 && py == 6) PG_MUL(SQ26);
    else if (h->pg_kind == 1 && px == 2 && py == 4) PG_MUL(SQ24);
    else if (h->pg_kind == 1 && px == 2 && py == 10) PG_MUL(SQ2A);
    else if (h->pg_kind == 1 && px == 4 && py == 6) PG_MUL(SQ46);
    else if (h->pg_kind == 2 && px == 3 && py == 2) PG_MUL(HC32);
    else if (h->pg_kind == 2 && px == 4 && py == 2) PG_MUL(HC42);
    else if (h->pg_kind == 2 && px == 3 && py == 3) PG_MUL(HC33);
    else if (h->pg_kind == 3 && px == 2 && py == 2) PG_MUL(TR22);
    else if (h->pg_kind == 3 && px == 2 && py == 4) PG_MUL(TR24);
    else if (h->pg_kind == 3 && px == 2 && py == 6) PG_MUL(TR26);
    else if (h->pg_kind == 3 && px == 4 && py == 4) PG_MUL(TR44);
    else { elph_set_error("k_mul_pg: no instantiation for kind %d, %d x %d patches", h->pg_kind, px, py); return ELPH_E_UNSUPPORTED; }
#undef PG_MUL
    return pg_check("k_mul_pg");
}
