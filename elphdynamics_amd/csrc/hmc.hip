// hmc.hip — one HMC update of the Holstein model as a device-resident trajectory:
//   update!(model, hmc, fa, P)  ->  standard_update! (Nb = 1) / multitimestep_update! (Nb > 1),   HMC.jl:313-638
//   refresh_v!, refresh_ϕ!                                                                         HMC.jl:648-692
//   calc_H / calc_K / calc_S / calc_Sf                                                             HMC.jl:697-784
//   calc_Sb, calc_dSbdx!                                                                           PhononAction.jl:11-66,114-187
// x, v, ϕ±, Λϕ±, O⁻¹Λϕ±, dS/dx live on the device (layout S) for the whole trajectory and between trajectories; the host
// drives the leapfrog loop (every step ends in the CG stop test anyway), and only scalars (partial sums of S, K, the
// solver status) come back.  The random numbers the reference draws from model.rng are inputs of the call.
// No CPU fallback: every arithmetic step on lattice vectors is a kernel below or one of the solver's kernels.

#include <cmath>
#include <cstring>
#include <vector>

#include "elph_internal.h"

#define RC(call)                \
    do {                        \
        int _rc = (call);       \
        if (_rc) return _rc;    \
    } while (0)

namespace {

constexpr int TPB = 256;

struct HmcState {
    double *x = nullptr, *v = nullptr, *x0 = nullptr, *v0 = nullptr, *dS = nullptr, *y = nullptr;
    double *R2 = nullptr;        // [2 ndim] R±
    double *phi = nullptr;       // [2 ndim] ϕ±
    double *faM = nullptr;       // FourierAccelerator.M in layout S ([k][site])
    double *par = nullptr;       // [2N] ω, ω₄   (λ, λ₂, μ are h->d_lam)
    double *part = nullptr;      // [3 L] partial sums
    double dtau = 0.0;
    bool have_state = false;
};

// E(τ,s) = exp(-Δτ (λ x + λ₂ x² - μ)), x already in layout S  (update_model!, HolsteinModels.jl:526-549)
__global__ void __launch_bounds__(TPB) k_hmc_expV(double *__restrict__ E, const double *__restrict__ x,
                                                  const double *__restrict__ lam3, int N, long long n, double dtau) {
    const long long i = (long long)blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    const int s = (int)(i % N);
    const double xi = x[i];
    E[i] = exp(-dtau * (lam3[s] * xi + lam3[N + s] * (xi * xi) + -lam3[2 * N + s]));
}

// v = α v + sqrt(1-α²) y   (refresh_v!, HMC.jl:656)
__global__ void __launch_bounds__(TPB) k_hmc_refresh_v(double *__restrict__ v, const double *__restrict__ y, double alpha,
                                                       long long n) {
    const long long i = (long long)blockIdx.x * TPB + threadIdx.x;
    if (i < n) v[i] = alpha * v[i] + sqrt(1.0 - alpha * alpha) * y[i];
}

// v = v - cv Q;  x = x + cx v  (cx = 0: velocity half step only)   (HMC.jl:392-395,421)
__global__ void __launch_bounds__(TPB) k_hmc_leap(double *__restrict__ v, double *__restrict__ x, const double *__restrict__ Q,
                                                  double cv, double cx, long long n) {
    const long long i = (long long)blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    const double vn = v[i] - cv * Q[i];
    v[i] = vn;
    if (cx != 0.0) x[i] = x[i] + cx * vn;
}

// v = -v0  (rejected update, HMC.jl:453)
__global__ void __launch_bounds__(TPB) k_hmc_neg(double *__restrict__ v, const double *__restrict__ v0, long long n) {
    const long long i = (long long)blockIdx.x * TPB + threadIdx.x;
    if (i < n) v[i] = -v0[i];
}

// ϕ = Λ⁻¹ (MᵀR):  ϕ(τ) = -(1/Λ(τ)) u(τ-1),  ϕ(0) = +(1/Λ(0)) u(L-1)   (mulΛ⁻¹!, HMC.jl:978-995; Λ: :921-941)
__global__ void __launch_bounds__(TPB) k_hmc_phi(double *__restrict__ phi, const double *__restrict__ u,
                                                 const double *__restrict__ x, const double *__restrict__ lam3, int N, int L,
                                                 double dtau) {
    const long long n = (long long)N * L, i = (long long)blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    const int s = (int)(i % N), t = (int)(i / N);
    const int tm1 = (t == 0) ? L - 1 : t - 1;
    const double sg = (t == 0) ? 1.0 : -1.0;
    const double xi = x[i];
    const double Lam = exp(-dtau * (lam3[s] * xi + lam3[N + s] * (xi * xi)) / 2);
    const size_t o = (size_t)blockIdx.y * (size_t)n;
    phi[o + i] = sg * (1.0 / Lam) * u[o + (size_t)tm1 * N + s];
}

// dS/dx (+)= dSb/dx   (calc_dSbdx!, PhononAction.jl:114-187, no dispersive modes)
__global__ void __launch_bounds__(TPB) k_hmc_dsb(double *__restrict__ dS, const double *__restrict__ x,
                                                 const double *__restrict__ par, int N, int L, double dtau, int accumulate) {
    const long long n = (long long)N * L, i = (long long)blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    const int s = (int)(i % N), t = (int)(i / N);
    const int tp1 = (t == L - 1) ? 0 : t + 1, tm1 = (t == 0) ? L - 1 : t - 1;
    const double w = par[s], w4 = par[N + s];
    const double xt = x[i];
    double d = accumulate ? dS[i] : 0.0;
    d += (dtau * w * w) * xt;
    d += (dtau * 4 * w4) * xt * xt * xt;
    d -= (x[(size_t)tp1 * N + s] + x[(size_t)tm1 * N + s] - 2.0 * xt) / dtau;
    dS[i] = d;
}

__device__ __forceinline__ double blk_sum(double v, double *sc) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const int w = threadIdx.x >> 6, nw = TPB >> 6;
    if ((threadIdx.x & 63) == 0) sc[w] = v;
    __syncthreads();
    double t = 0.0;
    for (int k = 0; k < nw; ++k) t += sc[k];
    __syncthreads();
    return t;
}

// per-slice partial of Sb/Δτ   (calc_Sb, PhononAction.jl:11-66)
__global__ void __launch_bounds__(TPB) k_hmc_sb_part(double *__restrict__ part, const double *__restrict__ x,
                                                     const double *__restrict__ par, int N, int L, double dtau) {
    __shared__ double sc[8];
    const int t = blockIdx.x, tm1 = (t == 0) ? L - 1 : t - 1;
    double acc = 0.0;
    for (int s = threadIdx.x; s < N; s += TPB) {
        const double xt = x[(size_t)t * N + s], xm = x[(size_t)tm1 * N + s], w = par[s], w4 = par[N + s];
        acc += w * w * (xt * xt) / 2 + w4 * (xt * xt * xt * xt);
        acc += (xt - xm) * (xt - xm) / (dtau * dtau) / 2;
    }
    acc = blk_sum(acc, sc);
    if (threadIdx.x == 0) part[t] = acc;
}

// partial dot products: part[b] = sum over block b's range of a·b
__global__ void __launch_bounds__(TPB) k_hmc_dot_part(double *__restrict__ part, const double *__restrict__ a,
                                                      const double *__restrict__ b, long long n, long long per_block) {
    __shared__ double sc[8];
    const long long lo = (long long)blockIdx.x * per_block, hi = (lo + per_block < n) ? lo + per_block : n;
    double acc = 0.0;
    for (long long i = lo + threadIdx.x; i < hi; i += TPB) acc += a[i] * b[i];
    acc = blk_sum(acc, sc);
    if (threadIdx.x == 0) part[blockIdx.x] = acc;
}

inline unsigned nblk(long long n) { return (unsigned)((n + TPB - 1) / TPB); }

int chk(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { elph_set_error("%s: %s", what, hipGetErrorString(e)); return ELPH_E_HIP; }
    return ELPH_OK;
}

int dot_host(elph_handle_s *h, HmcState *st, const double *a, const double *b, long long n, double *out) {
    const int nb = (int)h->L;
    const long long per = (n + nb - 1) / nb;
    hipLaunchKernelGGL(k_hmc_dot_part, dim3((unsigned)nb), dim3(TPB), 0, h->stream, st->part, a, b, n, per);
    RC(chk("k_hmc_dot_part"));
    std::vector<double> p((size_t)nb);
    HIPCHK(hipMemcpyAsync(p.data(), st->part, sizeof(double) * nb, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    double s = 0.0;
    for (double q : p) s += q;
    *out = s;
    return ELPH_OK;
}

int calc_Sb(elph_handle_s *h, HmcState *st, double *out) {
    const int L = (int)h->L;
    hipLaunchKernelGGL(k_hmc_sb_part, dim3((unsigned)L), dim3(TPB), 0, h->stream, st->part, st->x, st->par, (int)h->N, L, st->dtau);
    RC(chk("k_hmc_sb_part"));
    std::vector<double> p((size_t)L);
    HIPCHK(hipMemcpyAsync(p.data(), st->part, sizeof(double) * L, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    double s = 0.0;
    for (double q : p) s += q;
    *out = st->dtau * s;
    return ELPH_OK;
}

int update_model(elph_handle_s *h, HmcState *st) {
    hipLaunchKernelGGL(k_hmc_expV, dim3(nblk(h->ndim)), dim3(TPB), 0, h->stream, h->d_E, st->x, h->d_lam, (int)h->N,
                       (long long)h->ndim, st->dtau);
    h->have_E = true;
    return chk("k_hmc_expV");
}

// calc_O⁻¹Λϕ!(hmc, model, P, power)  (HMC.jl:820-915): setup!(P), Λϕ±, both solves as one batch, iters = cld(total, 2)
int calc_OinvLphi(elph_handle_s *h, HmcState *st, int use_precond, double power, const double *kpm_randn, int64_t *kpm_calls,
                  int64_t *iters, int *flag) {
    const size_t nd = (size_t)h->ndim;
    int use = 0;
    if (use_precond) {
        const double *bmax = kpm_randn + (size_t)(2 * *kpm_calls) * (size_t)h->N, *bmin = bmax + h->N;
        ++*kpm_calls;
        int act = 0;
        RC(elph_kpm_setup(h, bmax, bmin, NAN, NAN, &act, nullptr, nullptr));
        use = 1;                      // an inactive preconditioner is the identity inside the preconditioned recurrence
    }
    RC(elph_launch_lambda_rhs(h, h->d_b, st->phi, st->x, st->dtau));
    HIPCHK(hipMemsetAsync(h->d_x, 0, 2 * nd * sizeof(double), h->stream));
    const double tol0 = h->tol;
    h->tol = pow(tol0, power);
    int64_t it2[2] = {0, 0};
    double res2[2];
    int fl2[2] = {0, 0};
    const int rc = elph_i_ldiv_core(h, 2, use, 0, it2, res2, fl2);
    h->tol = tol0;
    if (rc) return rc;
    int64_t tot = it2[0];
    int fl = fl2[0];
    if (fl == 0) { tot += it2[1]; fl = fl2[1]; }
    if (fl == 0) tot = (tot + 1) / 2;
    *iters = tot;
    *flag = fl;
    return ELPH_OK;
}

int fa(elph_handle_s *h, HmcState *st, double *out, const double *in, double power) {
    return elph_launch_fft_accel(h, out, in, st->faM, power, h->N);
}

// calc_H (HMC.jl:697-705): S = Sf + Sb (:745-756,768-784), K = v·(M v)/2 (:711-719)
int calc_H(elph_handle_s *h, HmcState *st, double *H, double *S, double *K) {
    double sf = 0, sb = 0, k = 0;
    RC(dot_host(h, st, h->d_b, h->d_x, 2 * (long long)h->ndim, &sf));
    RC(calc_Sb(h, st, &sb));
    RC(fa(h, st, st->y, st->v, 1.0));
    RC(dot_host(h, st, st->v, st->y, (long long)h->ndim, &k));
    *S = sf / 2 + sb;
    *K = k / 2;
    *H = *S + *K;
    return ELPH_OK;
}

// dS/dx = dSf/dx [+ dSb/dx]; Q = M^-1 dS/dx in place  (HMC.jl:379-384)
int force(elph_handle_s *h, HmcState *st, bool with_Sb) {
    RC(elph_launch_force_holstein(h, st->dS, h->d_x, st->phi, st->x, st->dtau));
    if (with_Sb) {
        hipLaunchKernelGGL(k_hmc_dsb, dim3(nblk(h->ndim)), dim3(TPB), 0, h->stream, st->dS, st->x, st->par, (int)h->N, (int)h->L,
                           st->dtau, 1);
        RC(chk("k_hmc_dsb"));
    }
    return fa(h, st, st->dS, st->dS, -1.0);
}

int boson_force(elph_handle_s *h, HmcState *st) {
    hipLaunchKernelGGL(k_hmc_dsb, dim3(nblk(h->ndim)), dim3(TPB), 0, h->stream, st->dS, st->x, st->par, (int)h->N, (int)h->L,
                       st->dtau, 0);
    RC(chk("k_hmc_dsb"));
    return fa(h, st, st->dS, st->dS, -1.0);
}

int leap(elph_handle_s *h, HmcState *st, double cv, double cx) {
    hipLaunchKernelGGL(k_hmc_leap, dim3(nblk(h->ndim)), dim3(TPB), 0, h->stream, st->v, st->x, st->dS, cv, cx, (long long)h->ndim);
    return chk("k_hmc_leap");
}

HmcState *state_of(elph_handle_s *h) { return static_cast<HmcState *>(h->hmc); }

}  // namespace

void elph_hmc_free(elph_handle_s *h) {
    HmcState *st = state_of(h);
    if (!st) return;
    double *ptrs[] = {st->x, st->v, st->x0, st->v0, st->dS, st->y, st->R2, st->phi, st->faM, st->par, st->part};
    for (double *p : ptrs) if (p) (void)hipFree(p);
    delete st;
    h->hmc = nullptr;
}

#define CHECK_H(h)                                    \
    do {                                              \
        if (!(h)) {                                   \
            elph_set_error("null handle");            \
            return ELPH_E_ARG;                        \
        }                                             \
        HIPCHK(hipSetDevice((h)->device));            \
    } while (0)

extern "C" int elph_hmc_create(elph_handle h, const double *omega, const double *omega4, const double *lambda,
                               const double *lambda2, const double *mu, double dtau, const double *fa_mass) {
    CHECK_H(h);
    if (h->kind != ELPH_MODEL_HOLSTEIN) { elph_set_error("HMC trajectory: Holstein handles only"); return ELPH_E_UNSUPPORTED; }
    if (!omega || !omega4 || !lambda || !lambda2 || !mu || !fa_mass || !(dtau > 0.0)) { elph_set_error("bad argument"); return ELPH_E_ARG; }
    RC(elph_i_ensure_capacity(h, 2));
    if (h->nchains != 1) { h->nchains = 1; elph_i_drop_graphs(h); }
    elph_hmc_free(h);
    HmcState *st = new HmcState();
    h->hmc = st;
    const size_t nd = (size_t)h->ndim, N = (size_t)h->N;
    double **vecs[] = {&st->x, &st->v, &st->x0, &st->v0, &st->dS, &st->y, &st->faM};
    for (double **p : vecs) HIPCHK(hipMalloc((void **)p, nd * sizeof(double)));
    HIPCHK(hipMalloc((void **)&st->R2, 2 * nd * sizeof(double)));
    HIPCHK(hipMalloc((void **)&st->phi, 2 * nd * sizeof(double)));
    HIPCHK(hipMalloc((void **)&st->par, 2 * N * sizeof(double)));
    HIPCHK(hipMalloc((void **)&st->part, 3 * (size_t)h->L * sizeof(double)));
    st->dtau = dtau;
    HIPCHK(hipMemcpyAsync(st->par, omega, N * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(st->par + N, omega4, N * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_lam, lambda, N * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_lam + N, lambda2, N * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_lam + 2 * N, mu, N * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_stage_in, fa_mass, nd * sizeof(double), hipMemcpyHostToDevice, h->stream));
    RC(elph_launch_r2s(h, st->faM, h->d_stage_in, 1));
    HIPCHK(hipMemsetAsync(st->v, 0, nd * sizeof(double), h->stream));
    HIPCHK(hipMemsetAsync(st->x, 0, nd * sizeof(double), h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return ELPH_OK;
}

extern "C" int elph_hmc_set_state(elph_handle h, const double *x, const double *v) {
    CHECK_H(h);
    HmcState *st = state_of(h);
    if (!st) { elph_set_error("elph_hmc_create has not been called"); return ELPH_E_STATE; }
    const size_t bytes = (size_t)h->ndim * sizeof(double);
    if (x) {
        HIPCHK(hipMemcpyAsync(h->d_stage_in, x, bytes, hipMemcpyHostToDevice, h->stream));
        RC(elph_launch_r2s(h, st->x, h->d_stage_in, 1));
        HIPCHK(hipStreamSynchronize(h->stream));
        st->have_state = true;
    }
    if (v) {
        HIPCHK(hipMemcpyAsync(h->d_stage_in, v, bytes, hipMemcpyHostToDevice, h->stream));
        RC(elph_launch_r2s(h, st->v, h->d_stage_in, 1));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    return ELPH_OK;
}

extern "C" int elph_hmc_get_state(elph_handle h, double *x, double *v) {
    CHECK_H(h);
    HmcState *st = state_of(h);
    if (!st) { elph_set_error("elph_hmc_create has not been called"); return ELPH_E_STATE; }
    const size_t bytes = (size_t)h->ndim * sizeof(double);
    if (x) {
        RC(elph_launch_s2r(h, h->d_stage_out, st->x, 1));
        HIPCHK(hipMemcpyAsync(x, h->d_stage_out, bytes, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    if (v) {
        RC(elph_launch_s2r(h, h->d_stage_out, st->v, 1));
        HIPCHK(hipMemcpyAsync(v, h->d_stage_out, bytes, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    return ELPH_OK;
}

extern "C" int elph_hmc_update(elph_handle h, double dt, int64_t nt, int nb, double alpha, int use_precond, const double *R,
                               const double *Rp, const double *Rm, const double *kpm_randn, double u_accept, int *accepted,
                               double *iters_per_solve, double *energies, int *flag_out) {
    CHECK_H(h);
    HmcState *st = state_of(h);
    if (!st) { elph_set_error("elph_hmc_create has not been called"); return ELPH_E_STATE; }
    if (!st->have_state) { elph_set_error("elph_hmc_set_state(x) has not been called"); return ELPH_E_STATE; }
    if (!R || !Rp || !Rm || !accepted || nt < 0 || nb < 1 || !(alpha >= 0.0 && alpha < 1.0) || !(dt > 0.0)) {
        elph_set_error("bad argument");
        return ELPH_E_ARG;
    }
    if (use_precond && !kpm_randn) { elph_set_error("kpm_randn required with a preconditioner"); return ELPH_E_ARG; }
    if (use_precond && !h->kpm_created) { elph_set_error("elph_kpm_create has not been called"); return ELPH_E_STATE; }
    RC(elph_i_ensure_capacity(h, 2));
    if (h->nchains != 1) { h->nchains = 1; elph_i_drop_graphs(h); }
    const size_t nd = (size_t)h->ndim, bytes = nd * sizeof(double);
    const long long n = (long long)nd;
    const double dtp = dt / (double)nb;
    int64_t iters = 0, itrs = 0, kpm_calls = 0;
    int flag = 0;
    double H0 = 0, H1 = 0, S = 0, K = 0;

    RC(update_model(h, st));
    // refresh_v!  (HMC.jl:648-659)
    HIPCHK(hipMemcpyAsync(h->d_stage_in, R, bytes, hipMemcpyHostToDevice, h->stream));
    RC(elph_launch_r2s(h, st->y, h->d_stage_in, 1));
    RC(fa(h, st, st->y, st->y, -0.5));
    hipLaunchKernelGGL(k_hmc_refresh_v, dim3(nblk(n)), dim3(TPB), 0, h->stream, st->v, st->y, alpha, n);
    RC(chk("k_hmc_refresh_v"));
    HIPCHK(hipMemcpyAsync(st->x0, st->x, bytes, hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(st->v0, st->v, bytes, hipMemcpyDeviceToDevice, h->stream));
    // refresh_ϕ!  (HMC.jl:665-692): ϕ± = Λ⁻¹ Mᵀ R±
    HIPCHK(hipMemcpyAsync(h->d_stage_in, Rp, bytes, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_stage_in + nd, Rm, bytes, hipMemcpyHostToDevice, h->stream));
    RC(elph_launch_r2s(h, st->R2, h->d_stage_in, 2));
    RC(elph_launch_mul(h, 1, h->d_b, st->R2, 2));
    hipLaunchKernelGGL(k_hmc_phi, dim3(nblk(n), 2), dim3(TPB), 0, h->stream, st->phi, h->d_b, st->x, h->d_lam, (int)h->N, (int)h->L,
                       st->dtau);
    RC(chk("k_hmc_phi"));

    RC(calc_OinvLphi(h, st, use_precond, 2.0, kpm_randn, &kpm_calls, &itrs, &flag));
    if (nb == 1) iters = itrs;        // standard_update! :373.  multitimestep_update! :507 has "iters += iters": not counted
    if (flag == 0) {
        RC(calc_H(h, st, &H0, &S, &K));
        RC(force(h, st, nb == 1));
        for (int64_t t = 1; t <= nt; ++t) {
            if (nb == 1) {
                RC(leap(h, st, dt / 2, dt));                                   // :392-395
            } else {
                RC(leap(h, st, dt / 2, 0.0));                                  // :525
                RC(boson_force(h, st));                                        // :528-532
                for (int tp = 1; tp <= nb; ++tp) {                             // :535-552
                    RC(leap(h, st, dtp / 2, dtp));
                    RC(boson_force(h, st));
                    RC(leap(h, st, dtp / 2, 0.0));
                }
            }
            RC(update_model(h, st));
            RC(calc_OinvLphi(h, st, use_precond, 1.0, kpm_randn, &kpm_calls, &itrs, &flag));
            iters += itrs;
            if (flag > 0) break;                                               // :405-408
            RC(force(h, st, nb == 1));
            RC(leap(h, st, dt / 2, 0.0));                                      // :418
        }
    }
    double P = 0.0;
    if (flag == 0) {
        RC(calc_OinvLphi(h, st, use_precond, 2.0, kpm_randn, &kpm_calls, &itrs, &flag));
        iters += itrs;
        if (flag == 0) {
            RC(calc_H(h, st, &H1, &S, &K));
            const double e = exp(-(H1 - H0));
            P = (1.0 < e) ? 1.0 : e;                                           // min(1, exp(-ΔH)); NaN -> NaN -> reject, like Julia
        }
    }
    const int acc = (u_accept < P && flag == 0) ? 1 : 0;                       // :441
    if (!acc) {
        HIPCHK(hipMemcpyAsync(st->x, st->x0, bytes, hipMemcpyDeviceToDevice, h->stream));
        hipLaunchKernelGGL(k_hmc_neg, dim3(nblk(n)), dim3(TPB), 0, h->stream, st->v, st->v0, n);
        RC(chk("k_hmc_neg"));
        RC(update_model(h, st));
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    *accepted = acc;
    if (iters_per_solve) *iters_per_solve = (double)((iters + (nt + 2) - 1) / (nt + 2));   // T1(cld(iters, Nt+2))
    if (energies) { energies[0] = H0; energies[1] = H1; energies[2] = S; energies[3] = K; energies[4] = P; }
    if (flag_out) *flag_out = flag;
    return ELPH_OK;
}
