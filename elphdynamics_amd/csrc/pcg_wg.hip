// pcg_wg.hip — workgroup-resident KPM-PRECONDITIONED conjugate gradient: the whole solve of M^T M x = b with the tau-FFT (KPM)
// preconditioner (IterativeSolvers.jl:153-234 with ldiv!(z, P, r) of KPMPreconditioners.jl:426-481) in ONE launch, for one to
// eight right-hand sides of Holstein models on the 16 x 16 square lattice with uniform hopping (BASELINE config C; the
// reference's real call shape: the two pseudofermion solves of an HMC force evaluation).
//
// Why: the streaming form of that iteration is five kernels (k_cg_ap, k_cg_xr, forward transform, Chebyshev, inverse transform).
// For one or two right-hand sides their own work is ~24 us (19 of it the longest Chebyshev recursion, which is the reference's
// algorithm) and the five dispatch boundaries cost another ~16: 40 us per iteration.  Here a TEAM of workgroups keeps the whole
// iteration on the chip and hands the vectors from stage to stage through L2 with self-tagged flags instead of kernel boundaries:
//
//   G "CG" workgroups      own the time slices (2 per wave) exactly as in k_cg_wg: p (own + halo slices) and exp(-dtau V) in
//                          registers, x and r in LDS; z = M^T M p, ONE meeting for p.z, r.z, z.z, r.r (cg_wg.hip), alpha, the
//                          updates and the stop test; then they write the new residual (layout S) to memory and raise flag B;
//   H "helper" workgroups  (8 waves each) wait for B and run P^-1 r in three stages:
//        forward twisted tau-transform   one 16 x 16 output tile per wave on the matrix cores (the tile code of k_dft_mfma_1),
//                                        spectrum nu[omega][site] to memory, flag C;
//        Chebyshev recursion             per frequency on two waves (Re / Im), registers only (kpm_sq_dev.h, 2 x 2 patch layout),
//                                        frequencies dealt longest first over the helper workgroups; result in place, the
//                                        r.(P^-1 r) partial sums in frequency space (Parseval), record D;
//        inverse transform               tiles again, P^-1 r (layout S) to memory, flag E;
//   the CG workgroups wait for E, add the helpers' partial sums to rho' = r.P^-1 r, and form p = P^-1 r + beta p from memory
//   (own and halo slices alike: no boundary exchange between CG workgroups is needed in this form).
//
// Every hand-over is "form R1" of the guide (Guideline 16): data by write-through (sc1) stores, every storing wave drains vmcnt,
// workgroup barrier, ONE tagged flag store per workgroup; consumers poll the producers' flags (one wave per workgroup), then read
// the data with sc1 loads.  Tags are the iteration number (+ a per-launch base): nothing is zeroed between iterations or launches.
// Buffers are reused safely: flag E of iteration k precedes every write of iteration k + 1 to r, nu and P^-1 r (see the stages).
// All sums are taken in fixed orders: a solve is bit-identical from run to run and independent of its companions in the batch.
// Against the streaming form it differs in summation trees only (same iteration counts up to the knife edge).

#include <algorithm>

#include "cg_wg_dev.h"
#include "kpm_sq_dev.h"

namespace wg {

typedef double double4_t __attribute__((ext_vector_type(4)));

// Diagnostic build (-DELPH_PCG_STAMPS, tools/time_pcg_phases.py): wave 0 of CG workgroup 0 and wave 0 of helper workgroup 0 (which
// carries the longest recursion) of right-hand side 0 add the wall-clock ticks (100 MHz) of their phases to a buffer.
#ifdef ELPH_PCG_STAMPS
__device__ unsigned long long g_pcg_stamps[32];
#define PST_DECL long long _pt = wall_clock64(); unsigned long long _pa[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define PST(k) do { __builtin_amdgcn_sched_barrier(0); const long long _n = wall_clock64(); _pa[k] += (unsigned long long)(_n - _pt); _pt = _n; __builtin_amdgcn_sched_barrier(0); } while (0)
#define PST_OUT(base, cond, iters) do { if (cond) { for (int _k = 0; _k < 12; ++_k) g_pcg_stamps[(base) + _k] = _pa[_k]; g_pcg_stamps[30] = (unsigned long long)(iters); } } while (0)
#else
#define PST_DECL
#define PST(k)
#define PST_OUT(base, cond, iters)
#endif

constexpr int PCG_H = 20;                    // helper workgroups per right-hand side (160 waves: one transform tile each at Ltau = 160)
constexpr int PCG_FLAGS = 32 + 32 + 64 + 32; // granules per right-hand side: flag B [32] | flag C [32] | record D [32][2] | flag E [32]

struct PcgCtl {
    u64 *flags;              // [nrhs][PCG_FLAGS]
    const double *Wf, *Wi;   // pre-swizzled A tiles of the forward / inverse twisted transform (dft_mfma.hip: k_dft_mfma_1's tables)
    int rtf, rti;            // row tiles of the forward (frequencies x re/im) and of the inverse (time slices) transform
    double2 *nu;             // [nrhs][Lo2][N] spectrum (complex)
    double *zp;              // [nrhs][L][N]   P^-1 r
    KpmDev K;
    const double *sqc, *sqs; // uniform hopping of the tau-averaged checkerboard: cosh, sinh
    int Lo2;
};

// ---- one 16 x 16 output tile of a tau-transform on the matrix cores (the arithmetic of k_dft_mfma_1; operands by sc1 loads, results by
// sc1 stores: producer and consumer sit in different workgroups of the same launch) -------------------------------------------------
template <int NT, bool INV>
__device__ __forceinline__ void dft_tile(double *__restrict__ out, const double *__restrict__ in, const double *__restrict__ W, int N,
                                         int L, int K, int mt, int ctile, int lane) {
    const int col = lane & 15, jj = lane >> 4;
    const int s = ctile * 16 + col;
    int sc = (s < N) ? s : N - 1;
    // (every operand address of a tile is loop-invariant over the iterations of the solve: hoisted, the 2 NT addresses spill and come
    //  back from scratch one by one in front of the MFMAs.  Laundering the lane's column keeps the address arithmetic — a few adds —
    //  inside the call; offsets are 32-bit so that a load is scalar base + vector offset)
    asm volatile("" : "+v"(sc));
    double a[NT], b[NT];
    const double *Wg = W + ((size_t)mt * NT) * WAVE + lane;
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) a[tt] = Wg[(size_t)tt * WAVE];
    if (!INV) {
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
            int t = 4 * tt + jj;
            t = (t < L) ? t : L - 1;
            b[tt] = ld_sc1(in + (unsigned)(t * N + sc));
        }
    } else {
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
            int k = 2 * tt + (jj >> 1);
            k = (k < K) ? k : K - 1;
            b[tt] = ld_sc1(in + (unsigned)((k * N + sc) * 2 + (jj & 1)));
        }
    }
    double4_t acc = (double4_t){0.0, 0.0, 0.0, 0.0}, acc1 = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int tt = 0; tt + 1 < NT; tt += 2) {
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[tt], b[tt], acc, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[tt + 1], b[tt + 1], acc1, 0, 0, 0);
    }
    if (NT & 1) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[NT - 1], b[NT - 1], acc, 0, 0, 0);
    acc += acc1;
    const int r0 = lane >> 4;
    if (!INV) {
        const int k = 8 * mt + 2 * r0;
        if (s < N) {
            if (k < K) { st_sc1(out + ((size_t)k * N + s) * 2, acc.x); st_sc1(out + ((size_t)k * N + s) * 2 + 1, acc.y); }
            if (k + 1 < K) { st_sc1(out + ((size_t)(k + 1) * N + s) * 2, acc.z); st_sc1(out + ((size_t)(k + 1) * N + s) * 2 + 1, acc.w); }
        }
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int t = 16 * mt + r0 + 4 * r;
            if (s < N && t < L) st_sc1(out + (size_t)t * N + s, acc[r]);
        }
    }
}

// one wave of a workgroup waits until the n flags at f carry `epoch` (payload in the low word: returned OR-ed); false = gave up / abort
__device__ __forceinline__ bool poll_flags(const u64 *f, int n, unsigned epoch, int lane, const WgCtl &R, unsigned &payload) {
    u64 v = 0;
    long long t_start = 0;
    for (int spin = 0;; ++spin) {
        bool ok = true;
        if (lane < n) { v = ld_gran(f + lane); ok = (unsigned)(v >> 32) == epoch; }
        if (__all(ok)) break;
        if (poll_bail<1>(spin, t_start, lane, R)) return false;
        __builtin_amdgcn_s_sleep(1);
    }
    const unsigned mine = (lane < n) ? (unsigned)v : 0u;
    payload = (__ballot(mine != 0u) != 0ull) ? 1u : 0u;       // (flags carry 0 or 1; records of sums are read by their consumer)
    return true;
}

template <int NT>
__global__ void __launch_bounds__(512) k_pcg_wg(CgBufs B, ModelDev m, WgCtl R, PcgCtl Pc) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int NPL = 4, T = 2, HS = NPL * WAVE;
    const int W = R.W, G = R.G, H = PCG_H;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & (WAVE - 1);
    const int rhs = blockIdx.x & 7, idx = blockIdx.x >> 3;          // one team per XCD (observed placement; speed only)
    if (rhs >= B.nrhs) return;
    const int N = m.N, L = m.L;
    const size_t ndim = (size_t)N * L;
    auto wrap = [L](int t) { return (t < 0) ? t + L : ((t >= L) ? t - L : t); };
    auto sgn = [](int t) { return (t == 0) ? -1.0 : 1.0; };
    u64 *flagB = Pc.flags + (size_t)rhs * PCG_FLAGS, *flagC = flagB + 32, *recD = flagB + 64, *flagE = flagB + 128;
    double *rg = B.r + (size_t)rhs * ndim, *zpg = Pc.zp + (size_t)rhs * ndim;
    double *nug = reinterpret_cast<double *>(Pc.nu + (size_t)rhs * Pc.Lo2 * N);
    const long long fixed_iters = R.fixed_iters;

    if (idx >= G) {
        // =====================================================================================================================
        // helper workgroup h: the three stages of P^-1 r, once per iteration
        // =====================================================================================================================
        const int h = idx - G;
        if (wv >= 8 || h >= H) return;
        double *xch = lds;                               // [4 pairs][2 parts][4 * 64]: Re/Im exchange of the two series of a frequency
        double *hsum = lds + 4 * 2 * HS;                 // [8] wave partial sums of r.(P^-1 r)
        double *hflag = hsum + 8;                        // [1] 1 run, 2 the solve has finished, 0 give up
        const int hw = h * 8 + wv;                       // helper wave of the team
        const int nct = (N + 15) / 16;
        const int ntf = Pc.rtf * nct, nti = Pc.rti * nct;
        // Chebyshev set-up (k_kpm_cheb_sq<2, UNI, ROWS>): the 2 x 2 patch layout, uniform hopping factored
        const KpmChainView V = kpm_chain_view(Pc.K, rhs, N);
        kpmsq::SqLane<2> Tq;
        Tq.yp = sq_patch_ycross(lane); Tq.ym = Tq.yp;
        int site[4];
        double eb[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { site[q] = sq_patch_site(lane, q); eb[q] = V.Ebar[site[q]]; }
        Tq.cu = Pc.sqc[0]; Tq.su = Pc.sqs[0];
        double ka = V.a;
        const double kb = V.b;
        {
            const double cc = Tq.cu * Tq.cu;
            Tq.su = Tq.su / Tq.cu; Tq.cu = cc * cc;
            ka *= Tq.cu;
        }
        const int pair = wv >> 1, part = wv & 1;
        const int nround = (Pc.Lo2 + 4 * H - 1) / (4 * H);
        double *xp = xch + (size_t)pair * 2 * HS;
        PST_DECL;
        for (long long seq = 0;; ++seq) {
            const unsigned epoch = R.epoch0 + (unsigned)seq + 1u;
            PST(11);
            // ---- stage a: the new residual of every CG workgroup is in memory (flag B; payload 1: the solve has ended) -------------
            if (wv == 0) {
                unsigned pay = 0;
                const bool ok = poll_flags(flagB, G, epoch, lane, R, pay);
                if (lane == 0) hflag[0] = !ok ? 0.0 : ((pay & 1u) ? 2.0 : 1.0);
            }
            wg_barrier();
            PST(0);
            if (hflag[0] != 1.0) { PST_OUT(12, rhs == 0 && h == 0 && wv == 0 && lane == 0, seq); return; }
            // ---- stage b: forward transform, one output tile per wave --------------------------------------------------------------
            // (the table pointers are laundered once per iteration: the tiles of W are loop-invariant, and hoisted out of the iteration
            //  loop their 2 x NT registers spill — reloaded from scratch one by one in front of the MFMAs that need them)
            const double *Wf = Pc.Wf, *Wi = Pc.Wi;
            asm volatile("" : "+s"(Wf), "+s"(Wi));
            for (int tix = hw; tix < ntf; tix += 8 * H) dft_tile<NT, false>(nug, rg, Wf, N, L, Pc.Lo2, tix / nct, tix % nct, lane);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            wg_barrier();
            PST(1);
            if (wv == 0) {
                if (lane == 0) st_gran(flagC + h, (u64)epoch << 32);
                unsigned pay = 0;
                const bool ok = poll_flags(flagC, H, epoch, lane, R, pay);
                if (!ok && lane == 0) hflag[0] = 0.0;
            }
            wg_barrier();
            PST(2);
            if (hflag[0] != 1.0) return;
            // ---- stage c: Chebyshev recursion per frequency (two waves: Re / Im), longest first over the helper workgroups -----------
            double dsum = 0.0;
            for (int rd = 0; rd < nround; ++rd) {
                const int wy = h + H * (pair + 4 * rd);
                const bool act = wy < Pc.Lo2;
                const int w = act ? V.wsched[wy] : 0;
                const int order = act ? V.order[w] : 1;
                const double2 *c = Pc.K.coeff + V.coff[w];
                double *u = nug + (size_t)w * N * 2;
                double vin[4], Pa[4], Qa[4], mid[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) vin[q] = act ? ld_sc1(u + 2 * site[q] + part) : 0.0;
                kpmsq::kpm_series_sq<2, true, true, true>(Pa, Qa, vin, eb, c, order, ka, kb, Tq);
#pragma unroll
                for (int q = 0; q < 4; ++q) xp[part * HS + q * WAVE + lane] = Qa[q];
                __syncthreads();
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const double Qo = xp[(part ^ 1) * HS + q * WAVE + lane];
                    mid[q] = (part == 0) ? Pa[q] + Qo : Pa[q] - Qo;
                }
                __syncthreads();
                kpmsq::kpm_series_sq<2, false, true, true>(Pa, Qa, mid, eb, c, order, ka, kb, Tq);
#pragma unroll
                for (int q = 0; q < 4; ++q) xp[part * HS + q * WAVE + lane] = Qa[q];
                __syncthreads();
                double dot = 0.0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const double Qo = xp[(part ^ 1) * HS + q * WAVE + lane];
                    const double res = (part == 0) ? Pa[q] - Qo : Pa[q] + Qo;
                    if (act) st_sc1(u + 2 * site[q] + part, res);
                    dot += vin[q] * res;
                }
                __syncthreads();
                // r.(P^-1 r) in frequency space (Parseval for the twisted transform; the mirror frequency contributes the same)
                dot = wave_sum_dpp(dot);
                const double wgt = ((L & 1) && w == Pc.Lo2 - 1) ? 1.0 : 2.0;
                if (act) dsum += wgt * dot / (double)L;
            }
            if (lane == 0) hsum[wv] = dsum;
            PST(3);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            wg_barrier();
            PST(4);
            if (wv == 0) {
                const double mine = wg_sum(hsum, 8, lane);
                if (lane < 2) {
                    const u64 bits = (u64)__double_as_longlong(mine);
                    st_gran(recD + 2 * h + lane, ((u64)epoch << 32) | (lane ? (bits >> 32) : (bits & 0xFFFFFFFFull)));
                }
                unsigned pay = 0;
                const bool ok = poll_flags(recD, 2 * H, epoch, lane, R, pay);
                if (!ok && lane == 0) hflag[0] = 0.0;
            }
            wg_barrier();
            PST(5);
            if (hflag[0] != 1.0) return;
            // ---- stage d: inverse transform ------------------------------------------------------------------------------------------
            for (int tix = hw; tix < nti; tix += 8 * H) dft_tile<NT, true>(zpg, nug, Wi, N, L, Pc.Lo2, tix / nct, tix % nct, lane);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            wg_barrier();
            PST(6);
            if (wv == 0 && lane == 0) st_gran(flagE + h, (u64)epoch << 32);
        }
    }

    // =========================================================================================================================
    // CG workgroup g: the time slices (k_cg_wg, DPP form, 2 slices per wave) with P^-1 r coming back from the helpers
    // =========================================================================================================================
    const int g = idx;
    if (wv >= W) return;
    const int t0 = (g * W + wv) * T;
    double *rall = lds;                                  // [W][T][HS]
    double *rl = rall + (size_t)wv * T * HS;
    double *xl = rall + (size_t)W * T * HS + (size_t)wv * T * HS;
    double *part = rall + (size_t)2 * W * T * HS, *tot = part + 32, *partF = part + 40;
    const CgParams P = B.params;
    CgState *st2 = B.state + 2 * rhs;
    const CgState S = ld_state(st2);
    int sc[NPL];
#pragma unroll
    for (int q = 0; q < NPL; ++q) sc[q] = sq_patch_site(lane, q);
    double *xg = B.x + (size_t)rhs * ndim;
    const double *p0g = B.p + (size_t)rhs * ndim;        // p0 = P^-1 r0 (parity 0 after elph_launch_cg_init)
    const double *Ech = m.E + (size_t)(rhs % m.nchains) * m.E_chain_stride;
    double zw[T + 1][NPL], p[T + 2][NPL], E[T + 1][NPL];
#pragma unroll
    for (int j = 0; j < T; ++j)
#pragma unroll
        for (int q = 0; q < NPL; ++q) {
            rl[j * HS + lane + q * WAVE] = rg[(size_t)(t0 + j) * N + sc[q]];
            xl[j * HS + lane + q * WAVE] = xg[(size_t)(t0 + j) * N + sc[q]];
        }
#pragma unroll
    for (int j = 0; j < T + 2; ++j)
#pragma unroll
        for (int q = 0; q < NPL; ++q) p[j][q] = p0g[(size_t)wrap(t0 + j - 1) * N + sc[q]];
#pragma unroll
    for (int j = 0; j < T + 1; ++j)
#pragma unroll
        for (int q = 0; q < NPL; ++q) E[j][q] = Ech[(size_t)wrap(t0 + j) * m.E_tau_stride + sc[q]];
    SqCtx<true> X;
    X.yx = sq_patch_ycross(lane);
    X.c[0][0] = m.c_uni; X.s[0][0] = m.s_uni / m.c_uni; X.k4 = (m.c_uni * m.c_uni) * (m.c_uni * m.c_uni);
    u64 *const slots0 = R.slots + (size_t)rhs * 2 * SLOTS_PER_RHS;      // (records by the parity of the iteration: cg_wg.hip)
    double rho = S.rho, kmin = S.kmin, eps = S.eps;
    const double eps0 = S.eps0, normb = S.normb;
    if (S.done || S.seq != 0) {                          // (the host guarantees a fresh solve; tell the helpers if not)
        if (wv == 0 && lane == 0) st_gran(flagB + g, ((u64)(R.epoch0 + 1u) << 32) | 1ull);
        return;
    }
    if (threadIdx.x == 0) tot[5] = 1.0;
    const double rr_far = (P.tol * normb) * (P.tol * normb) * 1.000001, y_num = 4.0 * (eps0 * normb) * (eps0 * normb);
    const double it_kappa = 0.17 * sqrt(P.kmax);
    PST_DECL;
    for (long long seq = 0;; ++seq) {
        const unsigned epoch = R.epoch0 + (unsigned)seq + 1u;
        u64 *const slotsA = slots0 + (size_t)(epoch & 1u) * SLOTS_PER_RHS, *const slotsB = slotsA + SLOTS_A;
        PST(11);
        // ---- z = M^T M p on the own slices (cg_wg.hip, DPP form) -----------------------------------------------------------------------
        double (&w)[T + 1][NPL] = zw;
#pragma unroll
        for (int k = 0; k <= T; ++k)
#pragma unroll
            for (int q = 0; q < 4; ++q) w[k][q] = E[k][q] * p[k][q];
        sq_sweepN<T + 1, false, true>(w, X);
#pragma unroll
        for (int k = 0; k <= T; ++k) {
            const double sg = sgn(wrap(t0 + k)) * X.k4;
#pragma unroll
            for (int q = 0; q < 4; ++q) w[k][q] = p[k + 1][q] - sg * w[k][q];
        }
        {
            double gq[T][4];
#pragma unroll
            for (int i = 0; i < T; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) gq[i][q] = w[i + 1][q];
            sq_sweepN<T, true, true>(gq, X);
#pragma unroll
            for (int i = 0; i < T; ++i) {
                const double sg = sgn(wrap(t0 + i + 1)) * X.k4;
#pragma unroll
                for (int q = 0; q < 4; ++q) w[i][q] = w[i][q] - sg * (E[i + 1][q] * gq[i][q]);      // z(t0 + i)
            }
        }
        double (&z)[T + 1][NPL] = zw;
        // ---- the meeting: p.z, r.z, z.z, r.r ---------------------------------------------------------------------------------------------
        double s_pz = 0.0, s_rz = 0.0, s_zz = 0.0, s_rr = 0.0;
#pragma unroll
        for (int j = 0; j < T; ++j)
#pragma unroll
            for (int q = 0; q < NPL; ++q) {
                const double rv = rl[j * HS + lane + q * WAVE];
                s_pz += p[j + 1][q] * z[j][q];
                s_rz += rv * z[j][q];
                s_zz += z[j][q] * z[j][q];
                s_rr += rv * rv;
            }
        {
            const double k4 = wave_sum4(s_pz, s_rz, s_zz, s_rr, lane);
            if (lane < 4) part[lane * 8 + wv] = k4;
        }
        PST(0);
        wg_barrier();
        PST(1);
        double pap, rz, zz, rr0;
        if (G == 1) {
            const double t4 = sum_part4(part, W, lane);
            pap = readlane_f64(t4, 0); rz = readlane_f64(t4, 8); zz = readlane_f64(t4, 16); rr0 = readlane_f64(t4, 24);
        } else {
            if (wv == 0) {
                publish_rec4(slotsA, g, sum_part4(part, W, lane), epoch, lane);
                double t4 = 0.0, d0, d1;
                bool ok;
                if (G <= 8) { u64 v[1] = {0}; ok = poll_rec4<1>(slotsA, G, nullptr, nullptr, epoch, lane, R, v, d0, d1); t4 = sum_rec4<1>(v, G, lane); }
                else        { u64 v[4] = {0, 0, 0, 0}; ok = poll_rec4<4>(slotsA, G, nullptr, nullptr, epoch, lane, R, v, d0, d1); t4 = sum_rec4<4>(v, G, lane); }
                if (lane < 8 && !(lane & 1)) tot[lane >> 1] = t4;
                if (!ok && lane == 0) tot[5] = 0.0;
            }
            wg_barrier();
            if (tot[5] == 0.0) return;
            pap = tot[0]; rz = tot[1]; zz = tot[2]; rr0 = tot[3];
        }
        PST(2);
        const double alpha = rho / pap;                                   // rho = r.(P^-1 r)   (:203-204)
        double rr = rr0 + alpha * (alpha * zz - 2.0 * rz);                // |r - alpha z|^2 by the one-step identity (cg_wg.hip)
#pragma unroll
        for (int j = 0; j < T; ++j)
#pragma unroll
            for (int q = 0; q < NPL; ++q) {
                const double rn = rl[j * HS + lane + q * WAVE] - alpha * z[j][q];
                rl[j * HS + lane + q * WAVE] = rn;
                xl[j * HS + lane + q * WAVE] += alpha * p[j + 1][q];
            }
        if (!(rr > 1e-3 * rr0)) {                                         // the identity cancels: sum the new residual itself
            double a = 0.0;
#pragma unroll
            for (int j = 0; j < T; ++j)
#pragma unroll
                for (int q = 0; q < NPL; ++q) { const double rn = rl[j * HS + lane + q * WAVE]; a += rn * rn; }
            a = wave_sum_dpp(a);
            if (lane == 0) partF[wv] = a;
            wg_barrier();
            if (G == 1) {
                rr = wg_sum(partF, W, lane);
            } else {
                if (wv == 0) {
                    const double mine = wg_sum(partF, W, lane);
                    if (lane < 2) {
                        const u64 bits = (u64)__double_as_longlong(mine);
                        st_gran(slotsB + 2 * g + lane, ((u64)epoch << 32) | (lane ? (bits >> 32) : (bits & 0xFFFFFFFFull)));
                    }
                    u64 v = 0;
                    const bool ok = poll_records(slotsB, G, epoch, lane, R, v);
                    const int half = (int)(unsigned)v;
                    double t = 0.0;
                    for (int k = 0; k < G; ++k)
                        t += __hiloint2double(__builtin_amdgcn_readlane(half, 2 * k + 1), __builtin_amdgcn_readlane(half, 2 * k));
                    if (lane == 0) { tot[4] = t; if (!ok) tot[5] = 0.0; }
                }
                wg_barrier();
                if (tot[5] == 0.0) return;
                rr = tot[4];
            }
        }
        // ---- stop test of iteration it = seq + 1 (IterativeSolvers.jl:211-219; screens as in cg_wg.hip) ----------------------------------
        const long long it = seq + 1;
        const bool fixed = fixed_iters > 0;
        int done = 0;
        const bool screened = !P.record_hist && it < (fixed ? fixed_iters : P.maxiter) && (fixed || rr > rr_far) &&
                              (rr + rr <= y_num || rr >= y_num + y_num) && (double)it < it_kappa;
        if (!screened) {
            eps = sqrt(rr) / normb;
            const double qq = 2.0 * (double)it / log(2.0 * eps0 / eps);
            const double val = qq * qq;
            kmin = (val > kmin) ? val : kmin;
            if (eps < P.tol) done = 1;
            else if (kmin > P.kmax) done = 2;
            else if (it >= P.maxiter) done = 3;
            if (fixed) done = (it >= fixed_iters) ? 3 : 0;
            if (g == 0 && wv == 0 && lane == 0 && P.record_hist) B.hist[(size_t)rhs * P.hist_stride + it] = eps;
        }
        PST(3);
        if (done) {
            PST_OUT(0, rhs == 0 && g == 0 && wv == 0 && lane == 0, it);
#pragma unroll
            for (int j = 0; j < T; ++j)
#pragma unroll
                for (int q = 0; q < NPL; ++q) xg[(size_t)(t0 + j) * N + sc[q]] = xl[j * HS + lane + q * WAVE];
            if (wv == 0 && lane == 0) {
                st_gran(flagB + g, ((u64)epoch << 32) | 1ull);           // the helpers leave
                if (g == 0) {
                    CgState o = S;
                    o.rho = rho; o.kmin = kmin; o.eps = eps; o.seq = it + 1; o.iters = it; o.done = done;
                    st2[0] = o;
                    st2[1] = o;
                }
            }
            return;
        }
        // ---- the new residual to memory for the transform; flag B ------------------------------------------------------------------------
#pragma unroll
        for (int j = 0; j < T; ++j)
#pragma unroll
            for (int q = 0; q < NPL; ++q) st_sc1(rg + (size_t)(t0 + j) * N + sc[q], rl[j * HS + lane + q * WAVE]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        wg_barrier();
        PST(4);
        // ---- wait for P^-1 r (flag E of every helper) and the partial sums of r.(P^-1 r) (records D) -----------------------------------------
        if (wv == 0) {
            if (lane == 0) st_gran(flagB + g, (u64)epoch << 32);
            unsigned pay = 0;
            bool ok = poll_flags(flagE, H, epoch, lane, R, pay);
            u64 v = 0;
            if (lane < 2 * H) v = ld_gran(recD + lane);                   // (complete since before flag E was raised)
            const int half = (int)(unsigned)v;
            double t = 0.0;
            for (int k = 0; k < H; ++k)
                t += __hiloint2double(__builtin_amdgcn_readlane(half, 2 * k + 1), __builtin_amdgcn_readlane(half, 2 * k));
            if (lane == 0) { tot[4] = t; if (!ok) tot[5] = 0.0; }
        }
        wg_barrier();
        PST(5);
        if (tot[5] == 0.0) return;
        const double rho1 = tot[4];
        const double beta = rho1 / rho;                                   // :221-224
        rho = rho1;
        // ---- p = P^-1 r + beta p on the own and the two halo slices, straight from memory ------------------------------------------------------
        double zt[T + 2][NPL];
#pragma unroll
        for (int j = 0; j < T + 2; ++j)
#pragma unroll
            for (int q = 0; q < NPL; ++q) zt[j][q] = ld_sc1(zpg + (size_t)wrap(t0 + j - 1) * N + sc[q]);
#pragma unroll
        for (int j = 0; j < T + 2; ++j)
#pragma unroll
            for (int q = 0; q < NPL; ++q) p[j][q] = zt[j][q] + beta * p[j][q];
        PST(6);
        wg_barrier();                                                     // (tot[4] is rewritten by the next iteration's fallback / wait)
    }
}

}  // namespace wg

#ifdef ELPH_PCG_STAMPS
extern "C" int elph_debug_pcg_stamps(unsigned long long *out32) {
    return hipMemcpyFromSymbol(out32, HIP_SYMBOL(wg::g_pcg_stamps), 32 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif

// Whether the resident preconditioned kernel takes this solve, and its team shape.
static bool pcg_shape(const elph_handle_s *h, int nrhs, int *Wo, int *Go, int *nto) {
    // Measured on MI355X (profiles/r03/pcg_wg_phase_stamps.log) this form takes 38.9 us per iteration for one
    // right-hand side of config C against 35.2 us of the five-kernel streaming form — the two tau-transforms cost 6 us each on the
    // 20 CUs of a team (two tiles per SIMD share the matrix core, and every stage ends in drain + barrier + flag + poll) where the
    // stand-alone kernels spread one tile per CU over 160 CUs and take ~4 us including their launch; the longest Chebyshev recursion
    // (16 us) is common to both forms.  Kept, tested (tests/test_gpu_parity.py) and timed (tools/time_pcg.py) as the measured answer
    // to "fold the preconditioned iteration into one launch".
    // Where it wins is 6..8 right-hand sides (eight teams on eight XCDs side by side: 39.9 us per iteration at 8 against 47.1 us
    // streaming, profiles/r03/time_kpm_streaming_small_batches.log) — the default; ELPH_PCG_WG=1 takes every batch of 1..8, =0 none.
    const char *eo = getenv("ELPH_PCG_WG");
    if (eo && eo[0] == '0') return false;
    if (!(eo && eo[0] == '1') && nrhs < 6) return false;
    if (!h->fast || h->wg_broken || h->kind != ELPH_MODEL_HOLSTEIN || h->sq_P != 2 || h->N != 256 || !h->sq_uniform || h->lp_mc != 4) return false;
    if (!h->kpm_ready || !h->kpm_active || h->dot_hi != 0 || h->solo_chain >= 0) return false;
    // (h->kpm_active says that SOME chain's expansion is active; the kernel runs the series of every right-hand side's chain without
    //  looking at KpmChainView::active, so a chain whose expansion is the identity — lam_mag uploaded as -1, tables unset — keeps the
    //  whole batch on the streaming form, which hands such a chain z = r)
    for (int c = 0; c < h->kpm_nch; ++c)
        if ((size_t)c < h->kpm_chain.size() ? !h->kpm_chain[(size_t)c].active : (h->h_lam.size() > 2 * (size_t)c + 1 && h->h_lam[2 * (size_t)c + 1] <= 0.0)) return false;
    if (nrhs < 1 || nrhs > 8) return false;
    const int L = (int)h->L;
    if (L % 2) return false;
    const int Wt = L / 2;
    int W = 0;
    for (int w = std::min(8, Wt); w >= 1; --w) if (Wt % w == 0) { W = w; break; }
    const int G = Wt / W;
    if (G + wg::PCG_H > 32 || (G > 1 && W < 2)) return false;
    const elph_handle_s::MfmaTab &Tf = h->mf[0][0], &Ti = h->mf[0][1];
    if (!Tf.W || !Ti.W || Tf.nt != Ti.nt || (Tf.nt != 20 && Tf.nt != 40)) return false;
    if (Wo) *Wo = W;
    if (Go) *Go = G;
    if (nto) *nto = Tf.nt;
    return true;
}

bool elph_pcg_wg_usable(const elph_handle_s *h, int nrhs) { return pcg_shape(h, nrhs, nullptr, nullptr, nullptr); }

// Runs the whole preconditioned CG for rhs [0, nrhs) after elph_launch_cg_init(h, nrhs, 1) (fixed_iters > 0: exactly that many
// iterations without stop test — measurement).  *ran = false: not applicable, nothing was launched.
int elph_pcg_wg(elph_handle_s *h, const CgBufs &B, int nrhs, long long fixed_iters, bool *ran) {
    *ran = false;
    int W = 0, G = 0, nt = 0;
    if (!B.params.use_prec || !pcg_shape(h, nrhs, &W, &G, &nt)) return ELPH_OK;
    ModelDev m = elph_model_dev(h);
    if (!m.uniform || !m.sq_bond) return ELPH_OK;
    const size_t n_slots = 2 * (size_t)nrhs * wg::SLOTS_PER_RHS, n_flags = (size_t)nrhs * wg::PCG_FLAGS;
    const size_t need = (n_slots + n_flags) * sizeof(wg::u64) + 64;
    const unsigned long long span = (unsigned long long)std::min<long long>(fixed_iters > 0 ? fixed_iters : B.params.maxiter, 1LL << 30) + 2;
    bool zero = false;
    if (need > h->res_cap) {
        HIPCHK(hipStreamSynchronize(h->stream));
        if (h->d_res) HIPCHK(hipFree(h->d_res));
        h->d_res = nullptr;
        HIPCHK(hipMalloc(&h->d_res, need));
        h->res_cap = need;
        zero = true;
    }
    if ((unsigned long long)h->wg_epoch + span >= 0xFFFFFFFFull) zero = true;
    if (zero) { HIPCHK(hipMemsetAsync(h->d_res, 0, h->res_cap, h->stream)); h->wg_epoch = 0; }
    wg::WgCtl R;
    char *base = static_cast<char *>(h->d_res);
    R.slots = reinterpret_cast<wg::u64 *>(base);
    R.bnd = nullptr;
    R.abort = reinterpret_cast<int *>(base + h->res_cap - 64);
    R.epoch0 = h->wg_epoch;
    h->wg_epoch += (unsigned)span;
    R.G = G; R.W = W;
    const char *eto = getenv("ELPH_WG_TIMEOUT_MS");
    R.timeout_ticks = (long long)(eto ? atoll(eto) : 2000) * 100000LL;
    R.fixed_iters = fixed_iters;
    R.x0_zero = 0;
    R.teams_per_xcd = 1;
    wg::PcgCtl Pc;
    Pc.flags = R.slots + n_slots;
    Pc.Wf = h->mf[0][0].W; Pc.Wi = h->mf[0][1].W;
    Pc.rtf = h->mf[0][0].groups * 5; Pc.rti = h->mf[0][1].groups * 5;       // (dft_mfma.hip: MG = 5 row tiles per group)
    Pc.nu = h->d_nu; Pc.zp = h->d_zp;
    Pc.K = elph_kpm_dev(h);
    Pc.sqc = h->d_sq_cbar; Pc.sqs = h->d_sq_sbar;
    Pc.Lo2 = (int)((h->L + 1) / 2);
    const size_t HS = 4 * WAVE;
    const size_t shm = std::max((size_t)2 * W * 2 * HS + 48, (size_t)4 * 2 * HS + 16) * sizeof(double);
    const dim3 grid((unsigned)(8 * (G + wg::PCG_H)));
    hipError_t e = hipSuccess;
    if (nt == 40) {
        e = hipFuncSetAttribute((const void *)wg::k_pcg_wg<40>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        if (e == hipSuccess) { hipLaunchKernelGGL((wg::k_pcg_wg<40>), grid, dim3(512), shm, h->stream, B, m, R, Pc); e = hipGetLastError(); }
    } else {
        e = hipFuncSetAttribute((const void *)wg::k_pcg_wg<20>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        if (e == hipSuccess) { hipLaunchKernelGGL((wg::k_pcg_wg<20>), grid, dim3(512), shm, h->stream, B, m, R, Pc); e = hipGetLastError(); }
    }
    if (e != hipSuccess) { elph_set_error("launch k_pcg_wg failed: %s", hipGetErrorString(e)); return ELPH_E_HIP; }
    h->wg_T = 2; h->wg_W = W; h->wg_G = G;
    h->wg_abort_off = h->res_cap - 64;
    *ran = true;
    return ELPH_OK;
}
