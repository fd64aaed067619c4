// kpm_dev.hip — the eigenvalue bounds of the KPM set-up on the device (KPMPreconditioners.jl:845-942,
// arnoldi_eigenvalue_bounds!): for every resident phonon configuration (chain) the n-step Arnoldi process on
// A = CBbar diag(Ebar) (largest Ritz value) and on A^-1 (1 / largest Ritz value), and the eigenvalues of the small
// Hessenberg matrices the reference hands to LAPACK's eigvals! — all chains at once, ONE wavefront per (chain, A | A^-1).
//
// Why a single wave: N <= 512 sites are 1..8 values per lane; a dot product is a few FMAs and a DPP reduction (no barrier, no
// atomics), the Krylov basis Q (n+1 vectors) sits in the wave's LDS, the checkerboard runs colour by colour on a private LDS
// slab exactly as in the solver kernels (a wave's DS operations retire in order).  The 210 sequential dot/axpy pairs of the
// modified Gram-Schmidt recursion — the reference's order of operations — cost ~30 us, the Hessenberg QR iteration
// (implicit double shift, the algorithm behind dhseqr; rows / columns of each reflector update spread over the lanes, matrix in
// LDS) about as much again; 128 such waves run side by side.  Round 1 did this on parked host threads: 1.1 ms per 64 chains.

#include <cmath>

#include "cg_fast_common.h"

namespace kd {

template <int CTRL>
__device__ __forceinline__ double dpp64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double wsum(double v) {
    v += dpp64<0xB1>(v);
    v += dpp64<0x4E>(v);
    v += dpp64<0x141>(v);
    v += dpp64<0x140>(v);
    const double r0 = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 0), __builtin_amdgcn_readlane(__double2loint(v), 0));
    const double r1 = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 16), __builtin_amdgcn_readlane(__double2loint(v), 16));
    const double r2 = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 32), __builtin_amdgcn_readlane(__double2loint(v), 32));
    const double r3 = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 48), __builtin_amdgcn_readlane(__double2loint(v), 48));
    return (r0 + r1) + (r2 + r3);
}

// the matrix / the slab belong to ONE wave and the LDS executes a wave's operations in order: what remains is that the compiler must not
// keep LDS values in registers across a step in which another lane rewrites them
#define LDS_ORDER() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

// Eigenvalues of the real upper-Hessenberg matrix a (column-major n x n, in LDS, n <= 64) by the implicit double-shift QR iteration
// (the algorithm behind dhseqr / EISPACK hqr).  One wave: the scalar control flow runs on values every lane reads from the same LDS
// words; the two SEARCHES of an iteration (small sub-diagonal from the bottom; start row of the double step) are evaluated by all
// lanes at once and decided by a ballot — walking them entry by entry costs an LDS round trip per row, 80 % of the whole kernel —
// and the two update loops of every reflector are spread over the lanes (one column / one row each).
// Returns the largest real part (the reference's maximum(real, eigvals!(h))), or +inf when the iteration does not converge.
__device__ double hess_max_real(double *a, int n, int lane) {
#define AH(i, j) a[(i) + (j) * n]
    auto sgn = [](double x, double y) { return (y >= 0.0) ? fabs(x) : -fabs(x); };
    if (n == 1) return AH(0, 0);
    double an = 0.0;
    for (int idx = lane; idx < n * n; idx += WAVE) { const int i = idx % n, j = idx / n; if (j >= i - 1) an += fabs(AH(i, j)); }
    const double anorm = wsum(an);
    double best = -INFINITY;
    int nn = n - 1;
    double t = 0.0, p = 0, q = 0, r = 0, s, x, y, z, w;
    while (nn >= 0) {
        int its = 0, l;
        do {
            {   // l = the largest row in [1, nn] whose sub-diagonal entry is negligible (0 if none): lane i tests row i
                bool hit = false;
                if (lane >= 1 && lane <= nn) {
                    double ss = fabs(AH(lane - 1, lane - 1)) + fabs(AH(lane, lane));
                    if (ss == 0.0) ss = anorm;
                    hit = (fabs(AH(lane, lane - 1)) + ss == ss);
                }
                const unsigned long long mk = __ballot(hit);
                l = mk ? 63 - __builtin_clzll(mk) : 0;
                if (l >= 1 && lane == 0) AH(l, l - 1) = 0.0;
                LDS_ORDER();
            }
            x = AH(nn, nn);
            if (l == nn) {                       // one root found
                best = fmax(best, x + t); --nn;
            } else {
                y = AH(nn - 1, nn - 1);
                w = AH(nn, nn - 1) * AH(nn - 1, nn);
                if (l == nn - 1) {               // two roots found
                    p = 0.5 * (y - x);
                    q = p * p + w;
                    z = sqrt(fabs(q));
                    x += t;
                    if (q >= 0.0) {
                        z = p + sgn(z, p);
                        double r1 = x + z, r2 = r1;
                        if (z != 0.0) r2 = x - w / z;
                        best = fmax(best, fmax(r1, r2));
                    } else {
                        best = fmax(best, x + p);    // complex pair: real part
                    }
                    nn -= 2;
                } else {                         // no roots yet: QR step
                    if (its == 60) return INFINITY;
                    if (its == 10 || its == 20) {   // exceptional shift
                        t += x;
                        for (int i = lane; i <= nn; i += WAVE) AH(i, i) = AH(i, i) - x;
                        LDS_ORDER();
                        s = fabs(AH(nn, nn - 1)) + fabs(AH(nn - 1, nn - 2));
                        y = x = 0.75 * s;
                        w = -0.4375 * s * s;
                    }
                    ++its;
                    // m = the largest row in [l, nn-2] at which the double step may start (two consecutive small sub-diagonal
                    // entries), m = l if none: lane i evaluates the test of row i
                    int m;
                    double pl = 0, ql = 0, rl = 0;
                    {
                        bool hit = false;
                        if (lane >= l && lane <= nn - 2) {
                            const double zz = AH(lane, lane);
                            double rr = x - zz, ss = y - zz;
                            pl = (rr * ss - w) / AH(lane + 1, lane) + AH(lane, lane + 1);
                            ql = AH(lane + 1, lane + 1) - zz - rr - ss;
                            rl = AH(lane + 2, lane + 1);
                            ss = fabs(pl) + fabs(ql) + fabs(rl);
                            pl /= ss; ql /= ss; rl /= ss;
                            if (lane == l) hit = true;
                            else {
                                const double u = fabs(AH(lane, lane - 1)) * (fabs(ql) + fabs(rl));
                                const double v = fabs(pl) * (fabs(AH(lane - 1, lane - 1)) + fabs(zz) + fabs(AH(lane + 1, lane + 1)));
                                hit = (u + v == v);
                            }
                        }
                        const unsigned long long mk = __ballot(hit);
                        m = 63 - __builtin_clzll(mk);          // lane l always hits
                        p = __shfl(pl, m, WAVE); q = __shfl(ql, m, WAVE); r = __shfl(rl, m, WAVE);
                    }
                    for (int i = m + 2 + lane; i <= nn; i += WAVE) {
                        AH(i, i - 2) = 0.0;
                        if (i != m + 2) AH(i, i - 3) = 0.0;
                    }
                    LDS_ORDER();
                    for (int k = m; k <= nn - 1; ++k) {
                        if (k != m) {
                            p = AH(k, k - 1); q = AH(k + 1, k - 1); r = (k != nn - 1) ? AH(k + 2, k - 1) : 0.0;
                            // (one reciprocal and three products where hqr divides three times — and likewise below: an f64 division is a dozen dependent
                            //  instructions, eight of them were half of a step of this latency-bound chain; the eigenvalues move by rounding only)
                            if ((x = fabs(p) + fabs(q) + fabs(r)) != 0.0) { const double ix = 1.0 / x; p *= ix; q *= ix; r *= ix; }
                        }
                        if ((s = sgn(sqrt(p * p + q * q + r * r), p)) != 0.0) {
                            if (lane == 0) {
                                if (k == m) {
                                    if (l != m) AH(k, k - 1) = -AH(k, k - 1);
                                } else {
                                    AH(k, k - 1) = -s * x;
                                }
                            }
                            p += s;
                            { const double is = 1.0 / s, ip = 1.0 / p; x = p * is; y = q * is; z = r * is; q *= ip; r *= ip; }
                            const bool three = (k != nn - 1);
                            for (int j = k + lane; j <= nn; j += WAVE) {            // rows k, k+1, k+2: one column per lane
                                const double a0 = AH(k, j), a1 = AH(k + 1, j), a2 = three ? AH(k + 2, j) : 0.0;
                                double pp = a0 + q * a1;
                                if (three) { pp += r * a2; AH(k + 2, j) = a2 - pp * z; }
                                AH(k + 1, j) = a1 - pp * y;
                                AH(k, j) = a0 - pp * x;
                            }
                            LDS_ORDER();
                            const int mmin = nn < k + 3 ? nn : k + 3;
                            for (int i = l + lane; i <= mmin; i += WAVE) {          // columns k, k+1, k+2: one row per lane
                                const double a0 = AH(i, k), a1 = AH(i, k + 1), a2 = three ? AH(i, k + 2) : 0.0;
                                double pp = x * a0 + y * a1;
                                if (three) { pp += z * a2; AH(i, k + 2) = a2 - pp * r; }
                                AH(i, k + 1) = a1 - pp * q;
                                AH(i, k) = a0 - pp;
                            }
                            LDS_ORDER();
                        }
                    }
                }
            }
        } while (l < nn - 1);
    }
    return best;
#undef AH
}

// A v = CBbar (Ebar .* v)  (KPMPreconditioners.jl:387-401)  or  A^-1 v = (CBbar^-1 v) ./ Ebar  (:406-420), in place on v[NPL] through the
// wave's LDS slab; the bond table is walked colour by colour (coloff: maximal runs of site-disjoint bonds), inverse: colours last to
// first with -s (Checkerboard.jl:298-316)
template <int NPL>
__device__ __forceinline__ void apply_A(double (&v)[NPL], double *slab, const double (&eb)[NPL], bool inverse, const int *bi, const int *bj,
                                        const int *coloff, int ncol, const double *cbar, const double *sbar, int lane) {
    if (!inverse) {
#pragma unroll
        for (int q = 0; q < NPL; ++q) slab[lane + q * WAVE] = eb[q] * v[q];
    } else {
#pragma unroll
        for (int q = 0; q < NPL; ++q) slab[lane + q * WAVE] = v[q];
    }
    LDS_ORDER();
    for (int cc = 0; cc < ncol; ++cc) {
        const int c = inverse ? ncol - 1 - cc : cc;
        for (int b = coloff[c] + lane; b < coloff[c + 1]; b += WAVE) {
            const int i = bi[b], j = bj[b];
            const double cb = cbar[b], sb = inverse ? -sbar[b] : sbar[b];
            const double t1 = slab[i], t2 = slab[j];
            slab[i] = cb * t1 + sb * t2;
            slab[j] = cb * t2 + sb * t1;
        }
        LDS_ORDER();
    }
#pragma unroll
    for (int q = 0; q < NPL; ++q) v[q] = inverse ? slab[lane + q * WAVE] / eb[q] : slab[lane + q * WAVE];
    LDS_ORDER();
}

// grid (nchains, 2): blockIdx.y = 0 largest Ritz value of A -> e_max, 1: of A^-1 -> e_min = 1 / it
template <int NPL>
__global__ void __launch_bounds__(WAVE) k_kpm_bounds(double *__restrict__ e_out /*[nch][2]: e_min, e_max*/, const double *__restrict__ Ebar,
                                                    const double *__restrict__ bstart /*[2][nch][N]: b_max, b_min*/, const int *__restrict__ bi,
                                                    const int *__restrict__ bj, const int *__restrict__ coloff, int ncol,
                                                    const double *__restrict__ cbar, const double *__restrict__ sbar, long long hop_stride,
                                                    int N, int n, int nch, int nb_lds) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int NS = NPL * WAVE;
    const int lane = threadIdx.x, chain = blockIdx.x;
    const bool inverse = (blockIdx.y == 1);
    double *Q = lds;                                  // [(n+1)][NS]
    double *slab = Q + (size_t)(n + 1) * NS;          // [NS]
    double *H = slab + NS;                            // [(n+1)][n] column-major with leading dimension n+1
    double *Awork = H + (size_t)(n + 1) * n;          // [n][n]
    const double *eb_g = Ebar + (size_t)chain * N;
    const double *cb = cbar + (size_t)chain * hop_stride, *sb = sbar + (size_t)chain * hop_stride;
    // the bond program of the checkerboard -> LDS (nb_lds = number of bonds when the host found room for it): an Arnoldi step walks it once,
    // colour by colour, a dependent chain of loads — from L2 that was most of a step
    if (nb_lds > 0) {
        double *lcb = Awork + (size_t)n * n, *lsb = lcb + nb_lds;
        int *lbi = reinterpret_cast<int *>(lsb + nb_lds), *lbj = lbi + nb_lds;
        for (int b = lane; b < nb_lds; b += WAVE) { lcb[b] = cb[b]; lsb[b] = sb[b]; lbi[b] = bi[b]; lbj[b] = bj[b]; }
        LDS_ORDER();
        cb = lcb; sb = lsb; bi = lbi; bj = lbj;
    }
    const double *b0 = bstart + ((size_t)(inverse ? 1 : 0) * nch + chain) * N;
    double eb[NPL], b[NPL], v[NPL];
    bool live[NPL];
#pragma unroll
    for (int q = 0; q < NPL; ++q) {
        const int s = lane + q * WAVE;
        live[q] = s < N;
        eb[q] = live[q] ? eb_g[s] : 1.0;
        b[q] = live[q] ? b0[s] : 0.0;
    }
    for (int i = lane; i < (n + 1) * n; i += WAVE) H[i] = 0.0;
    double nrm = 0.0;
#pragma unroll
    for (int q = 0; q < NPL; ++q) nrm += b[q] * b[q];
    nrm = sqrt(wsum(nrm));
#pragma unroll
    for (int q = 0; q < NPL; ++q) { b[q] = b[q] / nrm; Q[lane + q * WAVE] = b[q]; }
    LDS_ORDER();
    int l = n;
    for (int k = 0; k < n; ++k) {
#pragma unroll
        for (int q = 0; q < NPL; ++q) v[q] = b[q];
        apply_A<NPL>(v, slab, eb, inverse, bi, bj, coloff, ncol, cb, sb, lane);
#pragma unroll
        for (int q = 0; q < NPL; ++q) if (!live[q]) v[q] = 0.0;
        for (int j = 0; j <= k; ++j) {                               // modified Gram-Schmidt, the reference's order (:871-875)
            double qj[NPL], d = 0.0;
#pragma unroll
            for (int q = 0; q < NPL; ++q) { qj[q] = Q[(size_t)j * NS + lane + q * WAVE]; d += qj[q] * v[q]; }
            d = wsum(d);
            if (lane == 0) H[j + (size_t)(n + 1) * k] = d;
#pragma unroll
            for (int q = 0; q < NPL; ++q) v[q] -= d * qj[q];
        }
        double nv = 0.0;
#pragma unroll
        for (int q = 0; q < NPL; ++q) nv += v[q] * v[q];
        nv = sqrt(wsum(nv));
        if (lane == 0) H[(k + 1) + (size_t)(n + 1) * k] = nv;
        if (nv > 1e-12) {
#pragma unroll
            for (int q = 0; q < NPL; ++q) { b[q] = v[q] / nv; Q[(size_t)(k + 1) * NS + lane + q * WAVE] = b[q]; }
            LDS_ORDER();
        } else {
            l = k + 1;
            break;
        }
    }
    LDS_ORDER();
    bool finite = true;
    for (int idx = lane; idx < l * l; idx += WAVE) {
        const int i = idx % l, j = idx / l;
        const double hv = H[i + (size_t)(n + 1) * j];
        Awork[i + (size_t)l * j] = hv;
        finite = finite && isfinite(hv);
    }
    LDS_ORDER();
    double best = INFINITY;
#ifdef ELPH_KB_NOQR      // (timing experiment, wrong bounds: the Arnoldi process alone)
    if (__all(finite)) best = Awork[0] + 1.0;
#else
    if (__all(finite)) best = hess_max_real(Awork, l, lane);
#endif
    if (lane == 0) {
        if (!inverse) e_out[2 * chain + 1] = best;                                   // e_max (:890-895)
        else e_out[2 * chain + 0] = isfinite(best) ? 1.0 / best : -INFINITY;         // e_min (:934-939)
    }
}

}  // namespace kd

// Arnoldi bounds of the first nch resident chains on the device.  d_bstart: [2][nch][N] start vectors (b_max then b_min) on the device;
// d_eout: [nch][2] (e_min, e_max).  Returns ELPH_E_UNSUPPORTED when the lattice is beyond one wave (N > 512): the caller then
// takes the host path (kpm_host.cpp).
int elph_kpm_bounds_dev(elph_handle_s *h, int nch, const double *d_bstart, double *d_eout) {
    const int N = (int)h->N;
    if (N > 512) return ELPH_E_UNSUPPORTED;
    int n = h->kpm_n;
    if (n > N) n = N;
    if (n < 1) n = 1;
    if (n > 64) return ELPH_E_UNSUPPORTED;
    const int npl = (N + WAVE - 1) / WAVE;
    const size_t NS = (size_t)npl * WAVE;
    size_t shm = ((size_t)(n + 1) * NS + NS + (size_t)(n + 1) * n + (size_t)n * n + 8) * sizeof(double);
    if (shm > 160 * 1024) return ELPH_E_UNSUPPORTED;
    int nb_lds = 0;                                   // the bond program in LDS too, where it fits (64 KB of the 160 are plenty for 8 waves per CU)
    {
        const size_t extra = (size_t)h->nb * (2 * sizeof(double) + 2 * sizeof(int)) + 16;
        if (h->nb > 0 && shm + extra <= 64 * 1024) { shm += extra; nb_lds = (int)h->nb; }
    }
    const long long hop_stride = h->kpm_hop_per_chain ? (long long)h->nb : 0;
    const dim3 grid((unsigned)nch, 2), block(WAVE);
#define KB_LAUNCH(NPLV)                                                                                                              \
    {                                                                                                                                \
        hipError_t e = hipFuncSetAttribute((const void *)kd::k_kpm_bounds<NPLV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm); \
        if (e != hipSuccess) { elph_set_error("k_kpm_bounds: %s", hipGetErrorString(e)); return ELPH_E_HIP; }                         \
        hipLaunchKernelGGL((kd::k_kpm_bounds<NPLV>), grid, block, shm, h->stream, d_eout, h->d_Ebar, d_bstart, h->d_bi, h->d_bj,       \
                           h->d_coloff, h->ncol, h->d_cbar, h->d_sbar, hop_stride, N, n, nch, nb_lds);                                \
    }
    switch (npl) {
        case 1: KB_LAUNCH(1); break;
        case 2: KB_LAUNCH(2); break;
        case 3: KB_LAUNCH(3); break;
        case 4: KB_LAUNCH(4); break;
        case 5: KB_LAUNCH(5); break;
        case 6: KB_LAUNCH(6); break;
        case 7: KB_LAUNCH(7); break;
        default: KB_LAUNCH(8); break;
    }
#undef KB_LAUNCH
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { elph_set_error("launch k_kpm_bounds failed: %s", hipGetErrorString(e)); return ELPH_E_HIP; }
    return ELPH_OK;
}
