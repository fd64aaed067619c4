// greens.hip — stochastic Green's-function estimator on the device:
//   update!(estimator, model, P)      GreensFunctions.jl:201-234   n_v solves M⁻¹r = (MᵀM)⁻¹ Mᵀ r as ONE batched CG
//   setup!(estimator, n₁, n₂)         GreensFunctions.jl:239-288   the four translation-averaged products
//   convolve!, antiperiodic_copy!, periodic_product!               :351-440
//
// The reference doubles the time axis to 2L (antiperiodic copy [a, -a] or periodic copy [a·c, a·c]), runs 4-D FFTs
// (FFTW) of size 2L x L1 x L2 x L3 per orbital, multiplies fft(a)[ω,k]·fft(b)[-ω,-k]/V and transforms back: a circular
// cross-correlation  ab[Δ] = (1/V) Σ_x a~[x+Δ] b~[x].  Here the doubling is never materialised:
//   * an antiperiodic 2L-sequence has only odd frequencies and they are twice the L-point *twisted* spectrum
//     (TimeFreqFFTs.jl:55-73 — the transform the KPM preconditioner already owns); a periodic one only even
//     frequencies = twice the plain L-point spectrum; both inputs are real, so half spectra suffice and
//     fft(b)[-ω,-k] = conj fft(b)[ω,k];
//   * R and M⁻¹R sit in layout S (slice-major) after the solve, which is the layout the τ-DFT kernels of dft.hip read
//     with coalesced rows; the spatial transform of one frequency slice (N complex numbers) fits in LDS, so
//     forward spatial DFT of a and b, the orbital outer product, and the inverse spatial DFT are ONE kernel
//     (k_gr_spatial, one workgroup per (product, frequency));
//   * the result is real; the second half of the 2L axis is ∓ the first.  Output arrays keep the reference's shape
//     Complex[2L, n_s, n_s, L1, L2, L3] (imaginary parts are exact zeros) so measure_GΔ0 & co. index them unchanged.
// Spatial extents are the lattice's (8…32): direct DFTs with host-built twiddles, exact index reduction.

#include <cmath>
#include <vector>

#include "elph_internal.h"

#define RC(call)                \
    do {                        \
        int _rc = (call);       \
        if (_rc) return _rc;    \
    } while (0)

#define CHECK_H(h)                                                    \
    do {                                                              \
        if (!(h)) { elph_set_error("null handle"); return ELPH_E_ARG; } \
        HIPCHK(hipSetDevice((h)->device));                            \
    } while (0)

namespace {

constexpr int TPB = 256;

struct GreensState {
    int ns = 1, L1 = 1, L2 = 1, L3 = 1, nc = 1, nv = 2;
    bool have_vectors = false;
    double *R = nullptr, *X = nullptr;     // [nv][ndim] layout S: noise vectors and M⁻¹R
    double *f = nullptr;                   // [8][ndim] the eight real input fields of setup!
    double2 *nuA = nullptr;                // [2][Lo2][N] twisted half spectra of fields 0,1
    double2 *nuP = nullptr;                // [6][Lh][N]  plain half spectra of fields 2..7
    double2 *Y = nullptr;                  // [4][Kmax][ns*N] per-frequency spatial correlations
    double *C = nullptr;                   // [4][L][ns*N]    correlations, Δτ < L
    double2 *out = nullptr;                // [4][2L*ns*N]    reference layout
    double2 *tw = nullptr;                 // [L1 + L2 + L3] exp(-2πi j/Lx)
};

GreensState *gs_of(elph_handle_s *h) { return (GreensState *)h->greens; }

int gr_check(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { elph_set_error("launch %s failed: %s", what, hipGetErrorString(e)); return ELPH_E_HIP; }
    return ELPH_OK;
}

// The eight fields of setup! (GreensFunctions.jl:261-284), pointwise, layout S:
//   0: (M⁻¹r₁ + M⁻¹r₂)/√2   1: (r₁ + r₂)/√2            — antiperiodic pair (G[Δ,0])
//   2: M⁻¹r₁·M⁻¹r₂          3: r₁·r₂                   — G[Δ,0]·G[Δ,0]
//   4: M⁻¹r₂·r₂             5: M⁻¹r₁·r₁                — G[Δ,Δ]·G[0,0]
//   6: M⁻¹r₁·r₂             7: M⁻¹r₂·r₁                — G[Δ,0]·G[0,Δ]
__global__ void __launch_bounds__(TPB) k_gr_fields(double *__restrict__ f, const double *__restrict__ x1,
                                                   const double *__restrict__ x2, const double *__restrict__ r1,
                                                   const double *__restrict__ r2, long long n) {
    const long long i = (long long)blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    const double a1 = x1[i], a2 = x2[i], b1 = r1[i], b2 = r2[i];
    const double sq2 = sqrt(2.0);
    f[i] = (a1 + a2) / sq2;
    f[n + i] = (b1 + b2) / sq2;
    f[2 * n + i] = a1 * a2;
    f[3 * n + i] = b1 * b2;
    f[4 * n + i] = a2 * b2;
    f[5 * n + i] = a1 * b1;
    f[6 * n + i] = a1 * b2;
    f[7 * n + i] = a2 * b1;
}

__device__ __forceinline__ double2 cmul(double2 a, double2 b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

// dst[e] = Σ_j src[e with axis index j] · w^(a·j), a = axis index of e; w = tw (forward) or conj tw (inverse)
template <bool INV>
__device__ void dft_axis(double2 *dst, const double2 *src, int total, int stride, int len, const double2 *__restrict__ tw) {
    for (int e = threadIdx.x; e < total; e += TPB) {
        const int a = (e / stride) % len;
        const int base = e - a * stride;
        double2 acc = make_double2(0.0, 0.0);
        int m = 0;                                  // (a*j) mod len, exact
        for (int j = 0; j < len; ++j) {
            double2 w = tw[m];
            if (INV) w.y = -w.y;
            const double2 v = src[base + j * stride];
            acc.x += v.x * w.x - v.y * w.y;
            acc.y += v.x * w.y + v.y * w.x;
            m += a;
            if (m >= len) m -= len;
        }
        dst[e] = acc;
    }
    __syncthreads();
}

// DFT over the (up to) three cell axes of `total` = unit*L1*L2*L3 complex numbers held in LDS buffer a (scratch b);
// returns the buffer that holds the result.  unit = n_s when the orbital index is interleaved, 1 for cell arrays.
template <bool INV>
__device__ double2 *dft_cells(double2 *a, double2 *b, int unit, int L1, int L2, int L3, const double2 *__restrict__ tw) {
    const int total = unit * L1 * L2 * L3;
    if (L1 > 1) { dft_axis<INV>(b, a, total, unit, L1, tw); double2 *t = a; a = b; b = t; }
    if (L2 > 1) { dft_axis<INV>(b, a, total, unit * L1, L2, tw + L1); double2 *t = a; a = b; b = t; }
    if (L3 > 1) { dft_axis<INV>(b, a, total, unit * L1 * L2, L3, tw + L1 + L2); double2 *t = a; a = b; b = t; }
    return a;
}

// One workgroup per (frequency k, product c): spatial DFT of the two spectra, outer product over orbitals with
// fft(b)[-ω,-k] = conj fft(b)[ω,k], inverse spatial DFT.  Y[c][k][s2 + ns*(s1 + ns*cell)].
// LDS: 3 buffers of N complex + 2 of nc complex.
__global__ void __launch_bounds__(TPB) k_gr_spatial(double2 *__restrict__ Y, const double2 *__restrict__ nuA,
                                                    const double2 *__restrict__ nuB, int K, int N, int ns, int L1, int L2,
                                                    int L3, const double2 *__restrict__ tw, double norm,
                                                    long long conv_stride_in, long long conv_stride_out) {
    extern __shared__ double2 lds[];
    const int k = blockIdx.x, c = blockIdx.y, nc = L1 * L2 * L3;
    double2 *A = lds, *B = lds + N, *T = lds + 2 * N, *P = lds + 3 * N, *Q = P + nc;
    const double2 *a = nuA + (size_t)c * conv_stride_in + (size_t)k * N;
    const double2 *b = nuB + (size_t)c * conv_stride_in + (size_t)k * N;
    for (int e = threadIdx.x; e < N; e += TPB) A[e] = a[e];
    __syncthreads();
    double2 *Af = dft_cells<false>(A, T, ns, L1, L2, L3, tw);
    double2 *free1 = (Af == A) ? T : A;
    for (int e = threadIdx.x; e < N; e += TPB) B[e] = b[e];
    __syncthreads();
    double2 *Bf = dft_cells<false>(B, free1, ns, L1, L2, L3, tw);
    double2 *y = Y + (size_t)c * conv_stride_out + (size_t)k * ns * N;
    for (int s1 = 0; s1 < ns; ++s1) {
        for (int s2 = 0; s2 < ns; ++s2) {
            for (int q = threadIdx.x; q < nc; q += TPB) {
                const double2 av = Af[q * ns + s2], bv = Bf[q * ns + s1];
                P[q] = make_double2((av.x * bv.x + av.y * bv.y) * norm, (av.y * bv.x - av.x * bv.y) * norm);   // a·conj(b)
            }
            __syncthreads();
            double2 *Pf = dft_cells<true>(P, Q, 1, L1, L2, L3, tw);
            for (int q = threadIdx.x; q < nc; q += TPB) y[s2 + ns * (s1 + ns * q)] = Pf[q];
            __syncthreads();
        }
    }
}

// C[c][t][col] (real, Δτ < L) -> out[c][τ + 2L*col] complex for τ < 2L; the second half is sgn(c) times the first.
// 32x32 tile transpose through LDS: coalesced reads along col, coalesced writes along τ.
__global__ void __launch_bounds__(TPB) k_gr_out(double2 *__restrict__ out, const double *__restrict__ C, int L, int ncol) {
    __shared__ double tile[32][33];
    const int c = blockIdx.z;
    const double sgn = (c == 0) ? -1.0 : 1.0;
    const int col0 = blockIdx.x * 32, t0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
    const double *Cc = C + (size_t)c * L * ncol;
    for (int r = ty; r < 32; r += 8) {
        const int t = t0 + r, col = col0 + tx;
        tile[r][tx] = (t < L && col < ncol) ? Cc[(size_t)t * ncol + col] : 0.0;
    }
    __syncthreads();
    double2 *o = out + (size_t)c * 2 * L * ncol;
    for (int r = ty; r < 32; r += 8) {
        const int col = col0 + r, t = t0 + tx;
        if (t < L && col < ncol) {
            const double v = tile[tx][r];
            o[(size_t)col * 2 * L + t] = make_double2(v, 0.0);
            o[(size_t)col * 2 * L + L + t] = make_double2(sgn * v, 0.0);
        }
    }
}

template <class T>
int gr_alloc(T **p, size_t n) {
    if (*p) { HIPCHK(hipFree(*p)); *p = nullptr; }
    HIPCHK(hipMalloc((void **)p, std::max<size_t>(n, 1) * sizeof(T)));
    return ELPH_OK;
}

size_t spatial_lds_bytes(const elph_handle_s *h, const GreensState *g) { return (3 * (size_t)h->N + 2 * (size_t)g->nc) * sizeof(double2); }

}  // namespace

void elph_greens_free(elph_handle_s *h) {
    GreensState *g = gs_of(h);
    if (!g) return;
    void *ptrs[] = {g->R, g->X, g->f, g->nuA, g->nuP, g->Y, g->C, g->out, g->tw};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    delete g;
    h->greens = nullptr;
}

extern "C" int elph_greens_create(elph_handle h, int norbits, int L1, int L2, int L3, int nv) {
    CHECK_H(h);
    if (norbits < 1 || L1 < 1 || L2 < 1 || L3 < 1 || (int64_t)norbits * L1 * L2 * L3 != h->N) {
        elph_set_error("norbits*L1*L2*L3 = %lld does not match nsites = %lld", (long long)norbits * L1 * L2 * L3, (long long)h->N);
        return ELPH_E_ARG;
    }
    elph_greens_free(h);
    GreensState *g = new GreensState();
    h->greens = g;
    g->ns = norbits; g->L1 = L1; g->L2 = L2; g->L3 = L3; g->nc = L1 * L2 * L3;
    g->nv = std::max(2, nv);                                           // GreensFunctions.jl:167
    if (spatial_lds_bytes(h, g) > 160 * 1024) {
        elph_set_error("Green's-function estimator: a frequency slice of %lld sites does not fit in 160 KB of LDS", (long long)h->N);
        elph_greens_free(h);
        return ELPH_E_UNSUPPORTED;
    }
    const size_t nd = (size_t)h->ndim, N = (size_t)h->N, L = (size_t)h->L, Lo2 = (L + 1) / 2, Lh = L / 2 + 1;
    RC(gr_alloc(&g->R, (size_t)g->nv * nd));
    RC(gr_alloc(&g->X, (size_t)g->nv * nd));
    RC(gr_alloc(&g->f, 8 * nd));
    RC(gr_alloc(&g->nuA, 2 * Lo2 * N));
    RC(gr_alloc(&g->nuP, 6 * Lh * N));
    RC(gr_alloc(&g->Y, 4 * Lh * (size_t)g->ns * N));
    RC(gr_alloc(&g->C, 4 * L * (size_t)g->ns * N));
    RC(gr_alloc(&g->out, 4 * 2 * L * (size_t)g->ns * N));
    std::vector<double2> tw((size_t)L1 + L2 + L3);
    size_t o = 0;
    for (int len : {L1, L2, L3}) {
        for (int j = 0; j < len; ++j) {
            const double a = 2.0 * M_PI * (double)j / (double)len;
            tw[o + j] = make_double2(cos(a), -sin(a));
        }
        o += (size_t)len;
    }
    RC(gr_alloc(&g->tw, tw.size()));
    HIPCHK(hipMemcpy(g->tw, tw.data(), tw.size() * sizeof(double2), hipMemcpyHostToDevice));
    HIPCHK(hipFuncSetAttribute((const void *)k_gr_spatial, hipFuncAttributeMaxDynamicSharedMemorySize, (int)spatial_lds_bytes(h, g)));
    return ELPH_OK;
}

static int need_greens(elph_handle_s *h) {
    if (!h->greens) { elph_set_error("elph_greens_create has not been called"); return ELPH_E_STATE; }
    return ELPH_OK;
}

extern "C" int elph_greens_nv(elph_handle h, int *nv) {
    CHECK_H(h);
    RC(need_greens(h));
    if (nv) *nv = gs_of(h)->nv;
    return ELPH_OK;
}

// update!(estimator, model, P): the caller supplies the n_v noise vectors (the reference draws them from model.rng, :212)
extern "C" int elph_greens_update(elph_handle h, const double *R, int use_precond, int64_t *iters, double *residual_error,
                                  int *flag) {
    CHECK_H(h);
    RC(need_greens(h));
    if (!R) { elph_set_error("R is null"); return ELPH_E_ARG; }
    GreensState *g = gs_of(h);
    const int nv = g->nv;
    // several chains resident: right-hand side r of the batch belongs to chain r % nchains, so vector v of chain c is row
    // v * nchains + c of R — the estimator must then hold a multiple of nchains vectors
    if (h->nchains != 1 && nv % h->nchains) {
        elph_set_error("%d chains are resident: the estimator needs a multiple of that many vectors (it has %d)", h->nchains, nv);
        return ELPH_E_ARG;
    }
    const size_t nd = (size_t)h->ndim;
    RC(elph_i_ensure_capacity(h, nv));
    HIPCHK(hipMemcpyAsync(h->d_stage_in, R, (size_t)nv * nd * sizeof(double), hipMemcpyHostToDevice, h->stream));
    RC(elph_launch_r2s(h, g->R, h->d_stage_in, nv));
    RC(elph_launch_mul(h, 1, h->d_b, g->R, nv));                       // Mᵀr₁ (model.v″, :223-224)
    RC(elph_launch_zero(h, h->d_x, (int64_t)nv * h->ndim));            // fill!(M⁻¹r₁, 0)  :213
    std::vector<int64_t> it((size_t)nv);
    std::vector<double> res((size_t)nv);
    std::vector<int> fl((size_t)nv);
    RC(elph_i_ldiv_core(h, nv, use_precond ? 1 : 0, 0, it.data(), res.data(), fl.data()));
    HIPCHK(hipMemcpyAsync(g->X, h->d_x, (size_t)nv * nd * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    for (int i = 0; i < nv; ++i) {
        if (iters) iters[i] = it[i];
        if (residual_error) residual_error[i] = res[i];
        if (flag) flag[i] = fl[i];
    }
    g->have_vectors = true;
    return ELPH_OK;
}

// estimator.R / estimator.M⁻¹R as host arrays (n_v vectors of length ndim, reference layout); either may be NULL
extern "C" int elph_greens_set_vectors(elph_handle h, const double *R, const double *MinvR) {
    CHECK_H(h);
    RC(need_greens(h));
    GreensState *g = gs_of(h);
    const size_t bytes = (size_t)g->nv * (size_t)h->ndim * sizeof(double);
    RC(elph_i_ensure_capacity(h, g->nv));
    if (R) {
        HIPCHK(hipMemcpyAsync(h->d_stage_in, R, bytes, hipMemcpyHostToDevice, h->stream));
        RC(elph_launch_r2s(h, g->R, h->d_stage_in, g->nv));
    }
    if (MinvR) {
        HIPCHK(hipMemcpyAsync(h->d_stage_in, MinvR, bytes, hipMemcpyHostToDevice, h->stream));
        RC(elph_launch_r2s(h, g->X, h->d_stage_in, g->nv));
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    if (R && MinvR) g->have_vectors = true;
    return ELPH_OK;
}

extern "C" int elph_greens_get_vectors(elph_handle h, double *R, double *MinvR) {
    CHECK_H(h);
    RC(need_greens(h));
    GreensState *g = gs_of(h);
    if (!g->have_vectors) { elph_set_error("no vectors yet: call elph_greens_update or elph_greens_set_vectors"); return ELPH_E_STATE; }
    const size_t bytes = (size_t)g->nv * (size_t)h->ndim * sizeof(double);
    if (R) {
        RC(elph_launch_s2r(h, h->d_stage_out, g->R, g->nv));
        HIPCHK(hipMemcpyAsync(R, h->d_stage_out, bytes, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    if (MinvR) {
        RC(elph_launch_s2r(h, h->d_stage_out, g->X, g->nv));
        HIPCHK(hipMemcpyAsync(MinvR, h->d_stage_out, bytes, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    return ELPH_OK;
}

// setup!(estimator, n₁, n₂) — n₁, n₂ 1-based.  Each output (NULL = not copied back) is Complex[2L, n_s, n_s, L1, L2, L3],
// interleaved re/im, first index fastest.  The arrays also stay on the device (elph_greens_dev_arrays).
extern "C" int elph_greens_setup(elph_handle h, int n1, int n2, double *GD0, double *GD0_GD0, double *GDD_G00, double *GD0_G0D) {
    CHECK_H(h);
    RC(need_greens(h));
    GreensState *g = gs_of(h);
    if (!g->have_vectors) { elph_set_error("no vectors yet: call elph_greens_update or elph_greens_set_vectors"); return ELPH_E_STATE; }
    if (n1 < 1 || n1 > g->nv || n2 < 1 || n2 > g->nv) { elph_set_error("n1=%d, n2=%d outside 1..%d", n1, n2, g->nv); return ELPH_E_ARG; }
    const int N = (int)h->N, L = (int)h->L, Lo2 = (L + 1) / 2, Lh = L / 2 + 1, ns = g->ns, ncol = ns * N;
    const size_t nd = (size_t)h->ndim;
    const long long n = (long long)nd;
    hipLaunchKernelGGL(k_gr_fields, dim3((unsigned)((n + TPB - 1) / TPB)), dim3(TPB), 0, h->stream, g->f, g->X + (size_t)(n1 - 1) * nd,
                       g->X + (size_t)(n2 - 1) * nd, g->R + (size_t)(n1 - 1) * nd, g->R + (size_t)(n2 - 1) * nd, n);
    RC(gr_check("k_gr_fields"));
    RC(elph_dft_fwd_twisted(h, g->nuA, g->f, N, 2, nullptr));
    RC(elph_dft_fwd_plain(h, g->nuP, g->f + 2 * nd, N, 6));
    // total normalisation 1/(L·Nc)² (see header): 1/L comes from the inverse τ tables, the rest here
    const double norm = 1.0 / ((double)L * (double)g->nc * (double)g->nc);
    const size_t shm = spatial_lds_bytes(h, g);
    const long long ystride = (long long)Lh * ncol;
    hipLaunchKernelGGL(k_gr_spatial, dim3((unsigned)Lo2, 1), dim3(TPB), shm, h->stream, g->Y, g->nuA, g->nuA + (size_t)Lo2 * N, Lo2, N, ns,
                       g->L1, g->L2, g->L3, g->tw, norm, 0LL, 0LL);
    RC(gr_check("k_gr_spatial(twisted)"));
    hipLaunchKernelGGL(k_gr_spatial, dim3((unsigned)Lh, 3), dim3(TPB), shm, h->stream, g->Y + ystride, g->nuP, g->nuP + (size_t)Lh * N, Lh, N,
                       ns, g->L1, g->L2, g->L3, g->tw, norm, 2LL * Lh * N, ystride);
    RC(gr_check("k_gr_spatial(plain)"));
    RC(elph_dft_inv_twisted(h, g->C, g->Y, ncol, 1, nullptr, nullptr, nullptr, 0));
    // the plain inverse walks [rhs][Lh][ncol] spectra and writes [rhs][L][ncol]
    RC(elph_dft_inv_plain(h, g->C + (size_t)L * ncol, g->Y + ystride, ncol, 3));
    hipLaunchKernelGGL(k_gr_out, dim3((unsigned)((ncol + 31) / 32), (unsigned)((L + 31) / 32), 4), dim3(TPB), 0, h->stream, g->out, g->C, L,
                       ncol);
    RC(gr_check("k_gr_out"));
    const size_t cnt = 2 * (size_t)L * ncol;   // complex numbers per array
    // order of the device arrays: 0 GΔ0, 1 GΔ0·GΔ0, 2 GΔΔ·G00, 3 GΔ0·G0Δ (fields 0/1, 2/3, 4/5, 6/7)
    double *outs[4] = {GD0, GD0_GD0, GDD_G00, GD0_G0D};
    for (int c = 0; c < 4; ++c)
        if (outs[c]) HIPCHK(hipMemcpyAsync(outs[c], g->out + (size_t)c * cnt, cnt * sizeof(double2), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return ELPH_OK;
}

// Device-resident results of the last elph_greens_setup: arrays[c] (c = 0..3 as above), `count` complex numbers each.
extern "C" int elph_greens_dev_arrays(elph_handle h, void **arrays, int64_t *count) {
    CHECK_H(h);
    RC(need_greens(h));
    GreensState *g = gs_of(h);
    const size_t cnt = 2 * (size_t)h->L * (size_t)g->ns * (size_t)h->N;
    if (arrays) for (int c = 0; c < 4; ++c) arrays[c] = (void *)(g->out + (size_t)c * cnt);
    if (count) *count = (int64_t)cnt;
    return ELPH_OK;
}
