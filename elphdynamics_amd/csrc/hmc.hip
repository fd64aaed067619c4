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
#include <thread>
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
    int nch = 1;                 // chains advanced in lockstep (one phonon configuration, one trajectory each)
    bool ssh = false;            // SSH: the fields are bond phonons (nf = Nph columns), Λ ≡ 1
    int nf = 0;                  // phonon columns: nsites (Holstein) or Nph (SSH); field vectors are [tau][column]
    // per chain: [nch][nf*L], layout S (tau-major) inside a chain
    double *x = nullptr, *v = nullptr, *x0 = nullptr, *v0 = nullptr, *dS = nullptr, *y = nullptr;
    double *R2 = nullptr;        // [2][nch][ndim] R±
    double *phi = nullptr;       // [2][nch][ndim] ϕ±
    double *faM = nullptr;       // FourierAccelerator.M in layout S ([k][site]), shared by the chains
    double *par = nullptr;       // [2N] ω, ω₄   (λ, λ₂, μ are h->d_lam)
    double *part = nullptr;      // [2 nch][L] partial sums
    double dtau = 0.0;
    bool have_state = false;
    // SSH phonon types of the same name share their fields (SSHModels.jl:480-502): primary column of each column, the next
    // member of its class (-1 ends the list) and the weight 1 (primary) / 0 of each column in Sb and K
    bool shared = false;
    int *prim = nullptr, *next = nullptr;
    double *wcol = nullptr;
    std::vector<int> prim_host;
    // a sharded lattice with bond phonons (elph_shard_hmc_set_columns): the slab's phonon columns in the numbering of the whole lattice, which of
    // them this rank owns (weight 1 / 0, device copy in wown) — sums over fields count owned columns, the force of the others comes from their owners
    double *wown = nullptr;
    std::vector<int> gcol;
    std::vector<double> wown_host;
    int ngcol = 0;
    // optional generator for the random inputs the caller leaves NULL (elph_hmc_set_rng)
    bool rng_on = false;
    uint64_t rng_seed = 0, rng_batches = 0;
};

// The build's counter-based generator (elphdynamics_amd/synth.py): SplitMix64 -> uniform(0,1) -> Box-Muller.  Element i of a
// batch of n normals uses uniforms i/2 and ceil(n/2) + i/2; even i takes the cosine, odd i the sine.
__host__ __device__ inline uint64_t sm64(uint64_t seed, uint64_t idx1) {
    uint64_t z = seed + idx1 * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__host__ __device__ inline double u01(uint64_t seed, uint64_t i) {
    return ((double)(sm64(seed, i + 1) >> 11) + 0.5) * (1.0 / 9007199254740992.0);
}

// nvec vectors of ncols*L standard normals, written in layout S ([vec][tau][col]); the batch is numbered in the reference
// layout ([vec][col][tau]) so that it equals synth.randn(seed, nvec*ncols*L) handed over by the host
__global__ void __launch_bounds__(TPB) k_randn_S(double *__restrict__ dst, uint64_t seed, long long n, int ncols, int L) {
    const long long k = (long long)blockIdx.x * TPB + threadIdx.x;        // pair index
    const long long m = (n + 1) / 2;
    if (k >= m) return;
    const double r = sqrt(-2.0 * log(u01(seed, (uint64_t)k)));
    double sn, cs;
    sincos(6.283185307179586476925 * u01(seed, (uint64_t)(m + k)), &sn, &cs);
    const long long per = (long long)ncols * L;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const long long i = 2 * k + e;
        if (i >= n) break;
        const long long vec = i / per, w = i - vec * per;
        const int col = (int)(w / L), t = (int)(w - (long long)col * L);
        dst[vec * per + (long long)t * ncols + col] = r * (e ? sn : cs);
    }
}

// E(τ,s) = exp(-Δτ (λ x + λ₂ x² - μ)), x already in layout S  (update_model!, HolsteinModels.jl:526-549)
__global__ void __launch_bounds__(TPB) k_hmc_expV(double *__restrict__ E, const double *__restrict__ x,
                                                  const double *__restrict__ lam3, int N, long long n, double dtau,
                                                  const double *__restrict__ mu_ch, long long ndim) {
    const long long i = (long long)blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    const int s = (int)(i % N);
    const double xi = x[i];
    const double mu = mu_ch ? mu_ch[(i / ndim) * N + s] : lam3[2 * N + s];      // per chain when the tuner runs per chain
    E[i] = exp(-dtau * (lam3[s] * xi + lam3[N + s] * (xi * xi) + -mu));
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
                                                 double dtau, int nch) {
    // i runs over [chain][ndim]; blockIdx.y = sign; u, phi are [sign][chain][ndim]
    const long long nd = (long long)N * L, n = nd * nch, i = (long long)blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    const long long base = (i / nd) * nd;
    const int s = (int)(i % N), t = (int)((i % nd) / N);
    const int tm1 = (t == 0) ? L - 1 : t - 1;
    const double sg = (t == 0) ? 1.0 : -1.0;
    const double xi = x[i];
    const double Lam = exp(-dtau * (lam3[s] * xi + lam3[N + s] * (xi * xi)) / 2);
    const size_t o = (size_t)blockIdx.y * (size_t)n;
    phi[o + i] = sg * (1.0 / Lam) * u[o + (size_t)base + (size_t)tm1 * N + s];
}

// dS/dx (+)= dSb/dx   (calc_dSbdx!, PhononAction.jl:114-187, no dispersive modes)
__global__ void __launch_bounds__(TPB) k_hmc_dsb(double *__restrict__ dS, const double *__restrict__ x,
                                                 const double *__restrict__ par, int N, int L, double dtau, int accumulate,
                                                 int nch, const double *__restrict__ lam_shift = nullptr) {
    const long long nd = (long long)N * L, n = nd * nch, i = (long long)blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    const size_t base = (size_t)((i / nd) * nd);
    const int s = (int)(i % N), t = (int)((i % nd) / N);
    const int tp1 = (t == L - 1) ? 0 : t + 1, tm1 = (t == 0) ? L - 1 : t - 1;
    const double w = par[s], w4 = par[N + s];
    const double xt = x[i];
    double d = accumulate ? dS[i] : 0.0;
    d += (dtau * w * w) * xt;
    d += (dtau * 4 * w4) * xt * xt * xt;
    d -= (x[base + (size_t)tp1 * N + s] + x[base + (size_t)tm1 * N + s] - 2.0 * xt) / dtau;
    if (lam_shift) d -= dtau * lam_shift[s];         // calc_dSbdx!(…, shifted = true): the particle-hole shifted action (Langevin)
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
                                                     const double *__restrict__ par, int N, int L, double dtau,
                                                     const double *__restrict__ wcol, int col_lo = 0, int col_hi = 1 << 30) {
    __shared__ double sc[8];
    const int t = blockIdx.x, tm1 = (t == 0) ? L - 1 : t - 1;
    x += (size_t)blockIdx.y * (size_t)N * L;                   // blockIdx.y = chain
    part += (size_t)blockIdx.y * L;
    double acc = 0.0;
    for (int s = threadIdx.x; s < N; s += TPB) {
        if (s < col_lo || s >= col_hi) continue;               // a sharded lattice: this rank's own columns only
        if (wcol && wcol[s] == 0.0) continue;                  // only primary phonons count (PhononAction.jl:83)
        const double xt = x[(size_t)t * N + s], xm = x[(size_t)tm1 * N + s], w = par[s], w4 = par[N + s];
        acc += w * w * (xt * xt) / 2 + w4 * (xt * xt * xt * xt);
        acc += (xt - xm) * (xt - xm) / (dtau * dtau) / 2;
    }
    acc = blk_sum(acc, sc);
    if (threadIdx.x == 0) part[t] = acc;
}

// partial dot products of vector blockIdx.y (n elements each): part[y][b] = sum over block b's range of a·b
__global__ void __launch_bounds__(TPB) k_hmc_dot_part(double *__restrict__ part, const double *__restrict__ a,
                                                      const double *__restrict__ b, long long n, long long per_block,
                                                      int ncols = 0, int col_lo = 0, int col_hi = 0) {
    __shared__ double sc[8];
    a += (size_t)blockIdx.y * (size_t)n; b += (size_t)blockIdx.y * (size_t)n;
    part += (size_t)blockIdx.y * gridDim.x;
    const long long lo = (long long)blockIdx.x * per_block, hi = (lo + per_block < n) ? lo + per_block : n;
    double acc = 0.0;
    for (long long i = lo + threadIdx.x; i < hi; i += TPB) {
        if (ncols > 0) { const int c = (int)(i % ncols); if (c < col_lo || c >= col_hi) continue; }      // (sharded: own columns of [tau][col])
        acc += a[i] * b[i];
    }
    acc = blk_sum(acc, sc);
    if (threadIdx.x == 0) part[blockIdx.x] = acc;
}

inline unsigned nblk(long long n) { return (unsigned)((n + TPB - 1) / TPB); }

int chk(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { elph_set_error("%s: %s", what, hipGetErrorString(e)); return ELPH_E_HIP; }
    return ELPH_OK;
}

// out[k] = a_k · b_k for `count` consecutive vectors of n elements (count <= 2 nch)
// Shared fields (layout S: index = tau*nf + column).  k_alias_sum: every member of a class gets the class sum (muldMdx!,
// SSHModels.jl:820-826); k_alias_copy: every column takes its primary's value (randn!, :568-575); k_mask_cols: v *= wcol.
__global__ void __launch_bounds__(TPB) k_alias_sum(double *__restrict__ F, const int *__restrict__ prim, const int *__restrict__ next,
                                                   int nf, long long n) {
    const long long i = (long long)blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    const int col = (int)(i % nf);
    if (prim[col] != col || next[col] < 0) return;
    const long long base = i - col;
    double sum = F[i];
    for (int c = next[col]; c >= 0; c = next[c]) sum += F[base + c];
    F[i] = sum;
    for (int c = next[col]; c >= 0; c = next[c]) F[base + c] = sum;
}
__global__ void __launch_bounds__(TPB) k_alias_copy(double *__restrict__ F, const int *__restrict__ prim, int nf, long long n) {
    const long long i = (long long)blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    const int col = (int)(i % nf);
    if (prim[col] != col) F[i] = F[i - col + prim[col]];       // primaries are never written: no race
}
__global__ void __launch_bounds__(TPB) k_mask_cols(double *__restrict__ F, const double *__restrict__ wcol, int nf, long long n) {
    const long long i = (long long)blockIdx.x * TPB + threadIdx.x;
    if (i < n) F[i] *= wcol[i % nf];
}

// a sharded lattice (shard.hip: elph_shard_set_collectives): sums count this rank's own columns and are added over the ranks
bool sharded(const elph_handle_s *h) { return h->shard != nullptr && elph_i_shard_active(h); }

int dots_host(elph_handle_s *h, HmcState *st, const double *a, const double *b, long long n, int count, double *out, bool site_vectors = true) {
    const int nb = (int)h->L;
    const long long per = (n + nb - 1) / nb;
    int ncols = 0, clo = 0, chi = 0;
    // a shard: site vectors count the own rows; field vectors of the Holstein model are site vectors; bond-phonon fields were masked by the caller
    if (sharded(h) && (site_vectors || !st->wown)) { ncols = (int)(n / h->L); elph_i_shard_own_range(h, &clo, &chi); }
    hipLaunchKernelGGL(k_hmc_dot_part, dim3((unsigned)nb, (unsigned)count), dim3(TPB), 0, h->stream, st->part, a, b, n, per, ncols, clo, chi);
    RC(chk("k_hmc_dot_part"));
    std::vector<double> p((size_t)nb * count);
    HIPCHK(hipMemcpyAsync(p.data(), st->part, sizeof(double) * p.size(), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    for (int k = 0; k < count; ++k) {
        double s = 0.0;
        for (int q = 0; q < nb; ++q) s += p[(size_t)k * nb + q];
        out[k] = s;
    }
    if (sharded(h)) RC(elph_i_shard_allreduce(h, out, count));
    return ELPH_OK;
}

int alias_sum(elph_handle_s *h, HmcState *st, double *F) {
    if (!st->shared) return ELPH_OK;
    const long long n = (long long)st->nf * h->L * st->nch;
    hipLaunchKernelGGL(k_alias_sum, dim3(nblk(n)), dim3(TPB), 0, h->stream, F, (const int *)st->prim, (const int *)st->next, st->nf, n);
    return chk("k_alias_sum");
}

int calc_Sb(elph_handle_s *h, HmcState *st, double *out) {
    const int L = (int)h->L, nch = st->nch;
    int clo = 0, chi = 1 << 30;
    if (sharded(h) && !st->wown) elph_i_shard_own_range(h, &clo, &chi);
    const double *w = st->wown ? st->wown : (st->shared ? st->wcol : nullptr);      // (bond phonons on a shard: the owned columns)
    hipLaunchKernelGGL(k_hmc_sb_part, dim3((unsigned)L, (unsigned)nch), dim3(TPB), 0, h->stream, st->part, st->x, st->par, st->nf, L,
                       st->dtau, w, clo, chi);
    RC(chk("k_hmc_sb_part"));
    std::vector<double> p((size_t)L * nch);
    HIPCHK(hipMemcpyAsync(p.data(), st->part, sizeof(double) * p.size(), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    for (int c = 0; c < nch; ++c) {
        double s = 0.0;
        for (int q = 0; q < L; ++q) s += p[(size_t)c * L + q];
        out[c] = st->dtau * s;
    }
    if (sharded(h)) RC(elph_i_shard_allreduce(h, out, nch));
    return ELPH_OK;
}

int update_model(elph_handle_s *h, HmcState *st) {
    if (st->ssh) {      // SSHModels.jl:510-562 from the device-resident fields (tau-major)
        RC(elph_launch_ssh_update(h, st->x, st->nf, h->d_ssh_cb, h->d_ssh_par, h->d_ssh_tbare, h->d_ssh_slot, st->dtau, 1, st->nch));
        h->ssh_dtau = st->dtau;
        h->cs_host_stale = true;
        h->have_E = true;
        return ELPH_OK;
    }
    const long long n = (long long)h->ndim * st->nch;
    hipLaunchKernelGGL(k_hmc_expV, dim3(nblk(n)), dim3(TPB), 0, h->stream, h->d_E, st->x, h->d_lam, (int)h->N, n, st->dtau,
                       (const double *)((h->mu_per_chain && st->nch <= h->mu_ch_cap) ? h->d_mu_ch : nullptr), (long long)h->ndim);
    h->have_E = true;
    return chk("k_hmc_expV");
}

// calc_O⁻¹Λϕ!(hmc, model, P, power)  (HMC.jl:820-915) for every chain: setup!(P) per chain, Λϕ±, all 2·nch solves as one
// batch (right-hand side v·nch + c belongs to chain c), per chain iters = cld(total, 2) and flag
int calc_OinvLphi(elph_handle_s *h, HmcState *st, int use_precond, double power, const double *kpm_randn, int64_t *kpm_calls,
                  int64_t *iters, int *flag) {
    const size_t nd = (size_t)h->ndim;
    const int nch = st->nch;
    int use = 0;
    if (use_precond && sharded(h)) {
        // a sharded lattice: the expansion lives on the full-lattice handle (elph_shard_set_full_lattice), set up identically on every rank from
        // the SAME start vectors (kpm_randn holds vectors of the whole lattice) and the Ē summed over the ranks' own rows
        elph_handle_s *hf = elph_i_shard_full(h);
        const size_t NG = (size_t)hf->N;
        const double *bmax = kpm_randn + (size_t)(2 * *kpm_calls) * NG, *bmin = bmax + NG;
        ++*kpm_calls;
        if (st->ssh) {
            // bond phonons: the τ-means of cosh / sinh of every bond of the lattice, from the owners' slabs (update_A!, KPMPreconditioners.jl:355-381);
            // Λ ≡ 1 (HMC.jl:943-946,970-973): the right-hand sides are ϕ± themselves
            std::vector<double> cs;
            int64_t nbg = 0;
            RC(elph_i_shard_global_csbar(h, cs, &nbg));
            if (nbg != hf->nb) { elph_set_error("the full-lattice handle has %lld bonds, elph_shard_set_bonds named %lld", (long long)hf->nb, (long long)nbg); return ELPH_E_ARG; }
            RC(elph_i_kpm_setup_csbar(hf, cs.data(), cs.data() + nbg, bmax, bmin));
            HIPCHK(hipMemcpyAsync(h->d_b, st->phi, 2 * (size_t)nch * nd * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
        } else {
            std::vector<double> Eg;
            RC(elph_i_shard_global_ebar(h, Eg));
            RC(elph_i_kpm_setup_ebar(hf, Eg.data(), bmax, bmin));
            RC(elph_launch_lambda_rhs(h, h->d_b, st->phi, st->x, st->dtau, nch));
        }
        HIPCHK(hipMemsetAsync(h->d_x, 0, 2 * (size_t)nch * nd * sizeof(double), h->stream));
        h->x_zero = false;
        return elph_i_shard_solve_pair(h, hf, 1, power, iters, flag);
    }
    if (use_precond) {
        // start vectors of this set-up call: [b_max | b_min][chain][N]
        const double *bmax = kpm_randn + (size_t)(2 * *kpm_calls) * (size_t)nch * (size_t)h->N, *bmin = bmax + (size_t)nch * h->N;
        ++*kpm_calls;
        RC(elph_kpm_setup_chains(h, bmax, bmin, nullptr, nullptr, nullptr, nullptr, nullptr));
        use = 1;                      // an inactive preconditioner is the identity inside the preconditioned recurrence
    }
    if (st->ssh)        // Λ ≡ 1 (HMC.jl:943-946,970-973): the right-hand sides are ϕ± themselves
        HIPCHK(hipMemcpyAsync(h->d_b, st->phi, 2 * (size_t)nch * nd * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    else
        RC(elph_launch_lambda_rhs(h, h->d_b, st->phi, st->x, st->dtau, nch));
    HIPCHK(hipMemsetAsync(h->d_x, 0, 2 * (size_t)nch * nd * sizeof(double), h->stream));
    h->x_zero = true;
    if (sharded(h)) {      // one lattice over several ranks: the two solves one after the other through the sharded kernel (shard.hip)
        h->x_zero = false;
        return elph_i_shard_solve_pair(h, nullptr, 0, power, iters, flag);
    }
    const double tol0 = h->tol;
    h->tol = pow(tol0, power);
    std::vector<int64_t> it2((size_t)2 * nch, 0);
    std::vector<double> res2((size_t)2 * nch);
    std::vector<int> fl2((size_t)2 * nch, 0);
    const int rc = elph_i_ldiv_core(h, 2 * nch, use, 0, it2.data(), res2.data(), fl2.data());
    h->tol = tol0;
    if (rc) return rc;
    for (int c = 0; c < nch; ++c) {
        int64_t tot = it2[(size_t)c];
        int fl = fl2[(size_t)c];
        if (fl == 0) { tot += it2[(size_t)nch + c]; fl = fl2[(size_t)nch + c]; }
        if (fl == 0) tot = (tot + 1) / 2;
        iters[c] = tot;
        flag[c] = fl;
    }
    return ELPH_OK;
}

int fa(elph_handle_s *h, HmcState *st, double *out, const double *in, double power) {
    return elph_launch_fft_accel(h, out, in, st->faM, power, st->nf, st->nch);
}

// calc_H (HMC.jl:697-705) per chain: S = Sf + Sb (:745-756,768-784), K = v·(M v)/2 (:711-719)
int calc_H(elph_handle_s *h, HmcState *st, double *H, double *S, double *K) {
    const int nch = st->nch;
    std::vector<double> sf((size_t)2 * nch), sb((size_t)nch), k((size_t)nch);
    RC(dots_host(h, st, h->d_b, h->d_x, (long long)h->ndim, 2 * nch, sf.data()));
    RC(calc_Sb(h, st, sb.data()));
    RC(fa(h, st, st->y, st->v, 1.0));
    if (st->shared || st->wown) {       // calc_K of the SSH model counts primary fields only (HMC.jl:720-738); on a shard: the owned columns
        const long long nn = (long long)st->nf * h->L * nch;
        hipLaunchKernelGGL(k_mask_cols, dim3(nblk(nn)), dim3(TPB), 0, h->stream, st->y, (const double *)(st->wown ? st->wown : st->wcol), st->nf, nn);
        RC(chk("k_mask_cols"));
    }
    RC(dots_host(h, st, st->v, st->y, (long long)st->nf * h->L, nch, k.data(), false));
    for (int c = 0; c < nch; ++c) {
        S[c] = (sf[(size_t)c] + sf[(size_t)nch + c]) / 2 + sb[(size_t)c];
        K[c] = k[(size_t)c] / 2;
        H[c] = S[c] + K[c];
    }
    return ELPH_OK;
}

// dS/dx = dSf/dx [+ dSb/dx]; Q = M^-1 dS/dx in place  (HMC.jl:379-384)
int force(elph_handle_s *h, HmcState *st, bool with_Sb) {
    const long long n = (long long)st->nf * h->L * st->nch;
    if (st->ssh) {      // dSf/dx = -dMdx(M X₊, X₊) - dMdx(M X₋, X₋)  (HMC.jl:797-808; muldΛdx! is a no-op)
        RC(elph_launch_force_ssh(h, h->d_p, h->d_x, nullptr, st->nch));        // bond brackets q[chain][tau][bond] (d_p is free here)
        RC(elph_launch_ssh_scatter(h, st->dS, h->d_p, st->x, h->d_ssh_par, h->d_ssh_cb, st->nf, st->dtau, 1, -1.0, st->nch));
        RC(alias_sum(h, st, st->dS));
    } else {
        RC(elph_launch_force_holstein(h, st->dS, h->d_x, st->phi, st->x, st->dtau, st->nch));
    }
    // a sharded lattice: the fermion force is exact on the own rows (the MᵀM closure); the ghost rows — which the leapfrog moves along with
    // the own ones, so that the next update_model! sees the whole slab — take it from their owners.  The boson force and the Fourier
    // acceleration below are pointwise in the site index.
    if (sharded(h) && st->ssh) RC(elph_i_shard_ghost_sync_cols(h, st->dS, 1, st->nf, st->gcol.data(), st->ngcol, st->wown_host.data()));
    else if (sharded(h)) RC(elph_i_shard_ghost_sync(h, st->dS, 1));
    if (with_Sb) {
        hipLaunchKernelGGL(k_hmc_dsb, dim3(nblk(n)), dim3(TPB), 0, h->stream, st->dS, st->x, st->par, st->nf, (int)h->L, st->dtau, 1,
                           st->nch);
        RC(chk("k_hmc_dsb"));
    }
    return fa(h, st, st->dS, st->dS, -1.0);
}

int boson_force(elph_handle_s *h, HmcState *st) {
    const long long n = (long long)st->nf * h->L * st->nch;
    hipLaunchKernelGGL(k_hmc_dsb, dim3(nblk(n)), dim3(TPB), 0, h->stream, st->dS, st->x, st->par, st->nf, (int)h->L, st->dtau, 0,
                       st->nch);
    RC(chk("k_hmc_dsb"));
    return fa(h, st, st->dS, st->dS, -1.0);
}

int leap(elph_handle_s *h, HmcState *st, double cv, double cx) {
    const long long n = (long long)st->nf * h->L * st->nch;
    hipLaunchKernelGGL(k_hmc_leap, dim3(nblk(n)), dim3(TPB), 0, h->stream, st->v, st->x, st->dS, cv, cx, n);
    return chk("k_hmc_leap");
}

// host [nvec][ndim] (reference layout) -> device layout S, staged through the handle's staging buffer (cap = 2 nch vectors)
// ncols = 0: site vectors (ndim); otherwise field vectors with that many columns
int upload_vectors(elph_handle_s *h, double *dstS, const double *host, int nvec, int ncols = 0) {
    const size_t per = (size_t)(ncols > 0 ? ncols : h->N) * (size_t)h->L;
    HIPCHK(hipMemcpyAsync(h->d_stage_in, host, (size_t)nvec * per * sizeof(double), hipMemcpyHostToDevice, h->stream));
    return elph_launch_r2s(h, dstS, h->d_stage_in, nvec, ncols);
}

HmcState *state_of(elph_handle_s *h) { return static_cast<HmcState *>(h->hmc); }

// seed of the next batch: output number (batches drawn so far + 1) of SplitMix64 started at the handle's seed
uint64_t next_batch_seed(HmcState *st) { return sm64(st->rng_seed, ++st->rng_batches); }

// a random input vector set: uploaded when the caller gives it, otherwise drawn on the device (no PCIe, no host generator)
int randn_vectors(elph_handle_s *h, HmcState *st, double *dstS, const double *host, int nvec, int ncols = 0) {
    if (host) return upload_vectors(h, dstS, host, nvec, ncols);
    if (!st->rng_on) { elph_set_error("a random input is NULL and elph_hmc_set_rng has not been called"); return ELPH_E_ARG; }
    const int nc = ncols > 0 ? ncols : (int)h->N;
    const long long n = (long long)nvec * nc * h->L;
    hipLaunchKernelGGL(k_randn_S, dim3(nblk((n + 1) / 2)), dim3(TPB), 0, h->stream, dstS, next_batch_seed(st), n, nc, (int)h->L);
    RC(chk("k_randn_S"));
    if (ncols > 0 && st->shared) {      // randn!(v, ssh): v = v[primary_field]  (SSHModels.jl:568-575)
        hipLaunchKernelGGL(k_alias_copy, dim3(nblk(n)), dim3(TPB), 0, h->stream, dstS, (const int *)st->prim, nc, n);
        RC(chk("k_alias_copy"));
    }
    return ELPH_OK;
}

// host-side batches of the same generator: Arnoldi start vectors and Metropolis uniforms the caller leaves NULL
int randn_host(HmcState *st, std::vector<double> &out, size_t n) {
    if (!st->rng_on) { elph_set_error("a random input is NULL and elph_hmc_set_rng has not been called"); return ELPH_E_ARG; }
    const uint64_t seed = next_batch_seed(st);
    const size_t m = (n + 1) / 2;
    out.resize(n);
    auto fill = [&](size_t k0, size_t k1) {
        for (size_t k = k0; k < k1; ++k) {
            const double r = sqrt(-2.0 * log(u01(seed, k))), th = 6.283185307179586476925 * u01(seed, m + k);
            out[2 * k] = r * cos(th);
            if (2 * k + 1 < n) out[2 * k + 1] = r * sin(th);
        }
    };
    // counter-based: any split gives the same numbers.  ~20 ns per normal on one core; 64 chains x (Nt + 2) set-ups need 10^6
    const size_t nthr = std::min<size_t>({(size_t)16, m / 16384 + 1, (size_t)std::max(1u, std::thread::hardware_concurrency())});
    if (nthr <= 1) { fill(0, m); return ELPH_OK; }
    std::vector<std::thread> pool;
    for (size_t t = 0; t < nthr; ++t) pool.emplace_back(fill, m * t / nthr, m * (t + 1) / nthr);
    for (auto &th : pool) th.join();
    return ELPH_OK;
}
int uniform_host(HmcState *st, std::vector<double> &out, size_t n) {
    if (!st->rng_on) { elph_set_error("a random input is NULL and elph_hmc_set_rng has not been called"); return ELPH_E_ARG; }
    const uint64_t seed = next_batch_seed(st);
    out.resize(n);
    for (size_t k = 0; k < n; ++k) out[k] = u01(seed, k);
    return ELPH_OK;
}

}  // namespace

void elph_hmc_free(elph_handle_s *h) {
    HmcState *st = state_of(h);
    if (!st) return;
    double *ptrs[] = {st->x, st->v, st->x0, st->v0, st->dS, st->y, st->R2, st->phi, st->faM, st->par, st->part};
    for (double *p : ptrs) if (p) (void)hipFree(p);
    if (st->prim) (void)hipFree(st->prim);
    if (st->next) (void)hipFree(st->next);
    if (st->wcol) (void)hipFree(st->wcol);
    if (st->wown) (void)hipFree(st->wown);
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

static int hmc_create_core(elph_handle_s *h, int nchains, int nf, bool ssh, const double *omega, const double *omega4, double dtau,
                           const double *fa_mass) {
    // staging / scratch must hold the field vectors too (SSH: Nph columns may exceed the number of sites)
    const int per_field = (int)(((int64_t)nf + h->N - 1) / h->N);
    RC(elph_i_ensure_capacity(h, std::max(2 * nchains, 2 * per_field + 1)));
    elph_hmc_free(h);
    HmcState *st = new HmcState();
    h->hmc = st;
    st->nch = nchains; st->nf = nf; st->ssh = ssh;
    h->mu_per_chain = false;                 // a new dynamics state starts from the deck's one chemical potential
    const size_t nd = (size_t)h->ndim, nfd = (size_t)nf * (size_t)h->L, nc = (size_t)nchains;
    double **vecs[] = {&st->x, &st->v, &st->x0, &st->v0, &st->dS, &st->y};
    for (double **p : vecs) HIPCHK(hipMalloc((void **)p, nc * nfd * sizeof(double)));
    HIPCHK(hipMalloc((void **)&st->faM, nfd * sizeof(double)));
    HIPCHK(hipMalloc((void **)&st->R2, 2 * nc * nd * sizeof(double)));
    HIPCHK(hipMalloc((void **)&st->phi, 2 * nc * nd * sizeof(double)));
    HIPCHK(hipMalloc((void **)&st->par, 2 * (size_t)nf * sizeof(double)));
    HIPCHK(hipMalloc((void **)&st->part, 2 * nc * (size_t)h->L * sizeof(double)));
    st->dtau = dtau;
    HIPCHK(hipMemcpyAsync(st->par, omega, (size_t)nf * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(st->par + nf, omega4, (size_t)nf * sizeof(double), hipMemcpyHostToDevice, h->stream));
    RC(upload_vectors(h, st->faM, fa_mass, 1, nf));
    HIPCHK(hipMemsetAsync(st->v, 0, nc * nfd * sizeof(double), h->stream));
    HIPCHK(hipMemsetAsync(st->x, 0, nc * nfd * sizeof(double), h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return ELPH_OK;
}

extern "C" int elph_hmc_create_chains(elph_handle h, int nchains, const double *omega, const double *omega4, const double *lambda,
                                      const double *lambda2, const double *mu, double dtau, const double *fa_mass) {
    CHECK_H(h);
    if (h->kind != ELPH_MODEL_HOLSTEIN) { elph_set_error("elph_hmc_create[_chains]: Holstein handles (SSH: elph_hmc_create_ssh)"); return ELPH_E_UNSUPPORTED; }
    if (nchains < 1 || !omega || !omega4 || !lambda || !lambda2 || !mu || !fa_mass || !(dtau > 0.0)) { elph_set_error("bad argument"); return ELPH_E_ARG; }
    RC(elph_i_ensure_capacity(h, 2 * nchains));
    RC(elph_i_reserve_chains(h, nchains));
    const size_t N = (size_t)h->N;
    HIPCHK(hipMemcpyAsync(h->d_lam, lambda, N * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_lam + N, lambda2, N * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_lam + 2 * N, mu, N * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return hmc_create_core(h, nchains, (int)h->N, false, omega, omega4, dtau, fa_mass);
}

extern "C" int elph_hmc_create(elph_handle h, const double *omega, const double *omega4, const double *lambda,
                               const double *lambda2, const double *mu, double dtau, const double *fa_mass) {
    return elph_hmc_create_chains(h, 1, omega, omega4, lambda, lambda2, mu, dtau, fa_mass);
}

// HybridMonteCarlo for an SSH model (bond phonons): omega, omega4 per phonon (double[nph]), fa_mass double[nph * ltau]; the
// remaining arguments are those of elph_update_model_ssh_fields (couplings, checkerboard positions, bare hoppings, mu).
extern "C" int elph_hmc_create_ssh_chains(elph_handle h, int nchains, int64_t nph, const double *omega, const double *omega4,
                                          const int64_t *cb_index, const double *t_ph, const double *alpha, const double *alpha2,
                                          const double *t_bare_cb, const double *mu, double dtau, const double *fa_mass) {
    CHECK_H(h);
    if (h->kind != ELPH_MODEL_SSH) { elph_set_error("elph_hmc_create_ssh: SSH handles only"); return ELPH_E_UNSUPPORTED; }
    if (nchains < 1 || nph < 1 || !omega || !omega4 || !fa_mass || !(dtau > 0.0)) { elph_set_error("bad argument"); return ELPH_E_ARG; }
    if ((size_t)h->L * (size_t)h->nb > 2 * (size_t)h->ndim) { elph_set_error("more bonds than 2*nsites: scratch too small"); return ELPH_E_UNSUPPORTED; }
    RC(elph_i_reserve_chains(h, nchains));
    RC(elph_i_ssh_upload_params(h, nph, cb_index, t_ph, alpha, alpha2, t_bare_cb, mu));
    return hmc_create_core(h, nchains, (int)nph, true, omega, omega4, dtau, fa_mass);
}

extern "C" int elph_hmc_create_ssh(elph_handle h, int64_t nph, const double *omega, const double *omega4, const int64_t *cb_index,
                                   const double *t_ph, const double *alpha, const double *alpha2, const double *t_bare_cb,
                                   const double *mu, double dtau, const double *fa_mass) {
    return elph_hmc_create_ssh_chains(h, 1, nph, omega, omega4, cb_index, t_ph, alpha, alpha2, t_bare_cb, mu, dtau, fa_mass);
}

// x, v: double[nchains * ndim] (chain-major, reference layout inside a chain); NULL = leave
extern "C" int elph_hmc_set_state(elph_handle h, const double *x, const double *v) {
    CHECK_H(h);
    HmcState *st = state_of(h);
    if (!st) { elph_set_error("elph_hmc_create has not been called"); return ELPH_E_STATE; }
    if (x && st->shared) {      // update_model! refuses fields that differ from their primary (SSHModels.jl:549-559; isapprox)
        const size_t L = (size_t)h->L;
        for (int ch = 0; ch < st->nch; ++ch)
        for (int c = 0; c < st->nf; ++c) {
            const int pc = st->prim_host[(size_t)c];
            if (pc == c) continue;
            const size_t co = (size_t)ch * (size_t)st->nf * L;
            for (size_t t = 0; t < L; ++t) {
                const double a = x[co + (size_t)c * L + t], b = x[co + (size_t)pc * L + t];
                if (!(fabs(a - b) <= 1.4901161193847656e-08 * fmax(fabs(a), fabs(b)))) {
                    elph_set_error("(x[%zu]=%g) != (x[%zu]=%g): fields that share a primary field must be equal", (size_t)c * L + t + 1, a,
                                   (size_t)pc * L + t + 1, b);
                    return ELPH_E_ARG;
                }
            }
        }
    }
    if (x) {
        RC(upload_vectors(h, st->x, x, st->nch, st->nf));
        HIPCHK(hipStreamSynchronize(h->stream));
        st->have_state = true;
    }
    if (v) {
        RC(upload_vectors(h, st->v, v, st->nch, st->nf));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    return ELPH_OK;
}

extern "C" int elph_hmc_get_state(elph_handle h, double *x, double *v) {
    CHECK_H(h);
    HmcState *st = state_of(h);
    if (!st) { elph_set_error("elph_hmc_create has not been called"); return ELPH_E_STATE; }
    const size_t bytes = (size_t)st->nch * (size_t)st->nf * (size_t)h->L * sizeof(double);
    if (x) {
        RC(elph_launch_s2r(h, h->d_stage_out, st->x, st->nch, st->nf));
        HIPCHK(hipMemcpyAsync(x, h->d_stage_out, bytes, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    if (v) {
        RC(elph_launch_s2r(h, h->d_stage_out, st->v, st->nch, st->nf));
        HIPCHK(hipMemcpyAsync(v, h->d_stage_out, bytes, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    return ELPH_OK;
}

// One HMC update of every chain, in lockstep: the leapfrog schedule is common, each chain has its own field, momenta,
// pseudofermions, energies, Metropolis test and failure flag.  A chain whose solve fails (flag > 0, HMC.jl:405-408) is
// dead for this update: its field is put back to x0 at once (so that its remaining — ignored — solves stay cheap) and
// it is rejected at the end.
// model.μ changed (the chemical-potential tuner, MuFinder.jl:68-107: μ += Δμ on every site): refresh the device copy and the model
extern "C" int elph_hmc_set_mu(elph_handle h, const double *mu) {
    CHECK_H(h);
    HmcState *st = state_of(h);
    if (!st || !mu) { elph_set_error(st ? "null argument" : "elph_hmc_create / elph_langevin_create has not been called"); return st ? ELPH_E_ARG : ELPH_E_STATE; }
    const size_t N = (size_t)h->N;
    h->mu_per_chain = false;
    HIPCHK(hipMemcpyAsync(h->d_lam + (st->ssh ? 0 : 2 * N), mu, N * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (st->have_state) {
        RC(update_model(h, st));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    return ELPH_OK;
}

// the tuner with chains in lockstep: every chain its own chemical potential, mu[nchains][nsites]
extern "C" int elph_hmc_set_mu_chains(elph_handle h, const double *mu) {
    CHECK_H(h);
    HmcState *st = state_of(h);
    if (!st || !mu) { elph_set_error(st ? "null argument" : "elph_hmc_create / elph_langevin_create has not been called"); return st ? ELPH_E_ARG : ELPH_E_STATE; }
    const size_t N = (size_t)h->N, nch = (size_t)st->nch;
    HIPCHK(hipStreamSynchronize(h->stream));
    if ((int)nch > h->mu_ch_cap) {
        if (h->d_mu_ch) { HIPCHK(hipFree(h->d_mu_ch)); h->d_mu_ch = nullptr; }
        HIPCHK(hipMalloc((void **)&h->d_mu_ch, nch * N * sizeof(double)));
        h->mu_ch_cap = (int)nch;
    }
    HIPCHK(hipMemcpy(h->d_mu_ch, mu, nch * N * sizeof(double), hipMemcpyHostToDevice));
    h->mu_per_chain = true;
    if (st->have_state) {
        RC(update_model(h, st));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    return ELPH_OK;
}

extern "C" int elph_hmc_set_shared_fields(elph_handle h, const int64_t *primary_column) {
    CHECK_H(h);
    HmcState *st = state_of(h);
    if (!st || !st->ssh) { elph_set_error("elph_hmc_create_ssh / elph_langevin_create_ssh has not been called"); return ELPH_E_STATE; }
    if (!primary_column) { elph_set_error("bad argument"); return ELPH_E_ARG; }
    const int nf = st->nf;
    std::vector<int> prim((size_t)nf), next((size_t)nf, -1), last((size_t)nf);
    std::vector<double> w((size_t)nf);
    bool any = false;
    for (int c = 0; c < nf; ++c) {
        const int64_t p = primary_column[c];
        if (p < 0 || p > c || primary_column[p] != p) {
            elph_set_error("primary_column[%d] = %lld: a primary is an earlier (or the same) column that is its own primary", c, (long long)p);
            return ELPH_E_ARG;
        }
        prim[(size_t)c] = (int)p;
        last[(size_t)c] = c;
        w[(size_t)c] = (p == c) ? 1.0 : 0.0;
        if (p != c) {       // append to the class list of p, in column order
            next[(size_t)last[(size_t)p]] = c;
            last[(size_t)p] = c;
            any = true;
        }
    }
    if (!st->prim) {
        HIPCHK(hipMalloc((void **)&st->prim, (size_t)nf * sizeof(int)));
        HIPCHK(hipMalloc((void **)&st->next, (size_t)nf * sizeof(int)));
        HIPCHK(hipMalloc((void **)&st->wcol, (size_t)nf * sizeof(double)));
    }
    HIPCHK(hipMemcpy(st->prim, prim.data(), (size_t)nf * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(st->next, next.data(), (size_t)nf * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(st->wcol, w.data(), (size_t)nf * sizeof(double), hipMemcpyHostToDevice));
    st->prim_host = prim;
    st->shared = any;
    return ELPH_OK;
}

// Bond phonons on a sharded lattice: the slab's phonon columns (the phonons of its bonds, ghost bonds included) in the numbering of the whole
// lattice, and which of them this rank owns (the bond's first site lies in the own rows: every phonon has exactly one owner).  Call after
// elph_hmc_create_ssh on the slab handle.  Sums over fields (S_b, K) then count owned columns and are added over the ranks; the fermion
// force — exact on the owner, where the bond bracket is — reaches the other holders of a column through the host all-reduce.
extern "C" int elph_shard_hmc_set_columns(elph_handle h, const int64_t *global_column, int64_t n_global_columns, const double *own_weight) {
    CHECK_H(h);
    HmcState *st = state_of(h);
    if (!st || !st->ssh) { elph_set_error("elph_hmc_create_ssh has not been called"); return ELPH_E_STATE; }
    if (!h->shard) { elph_set_error("elph_shard_create has not been called"); return ELPH_E_STATE; }
    if (!global_column || !own_weight || n_global_columns < st->nf) { elph_set_error("bad argument"); return ELPH_E_ARG; }
    st->gcol.assign((size_t)st->nf, 0);
    st->wown_host.assign((size_t)st->nf, 0.0);
    for (int c = 0; c < st->nf; ++c) {
        if (global_column[c] < 0 || global_column[c] >= n_global_columns || !(own_weight[c] == 0.0 || own_weight[c] == 1.0)) {
            elph_set_error("column %d: global column %lld of %lld, weight %g", c, (long long)global_column[c], (long long)n_global_columns, own_weight[c]);
            return ELPH_E_ARG;
        }
        st->gcol[(size_t)c] = (int)global_column[c];
        st->wown_host[(size_t)c] = own_weight[c];
    }
    st->ngcol = (int)n_global_columns;
    if (!st->wown) HIPCHK(hipMalloc((void **)&st->wown, (size_t)st->nf * sizeof(double)));
    HIPCHK(hipMemcpy(st->wown, st->wown_host.data(), (size_t)st->nf * sizeof(double), hipMemcpyHostToDevice));
    return ELPH_OK;
}

extern "C" int elph_hmc_set_rng(elph_handle h, uint64_t seed) {
    CHECK_H(h);
    HmcState *st = state_of(h);
    if (!st) { elph_set_error("elph_hmc_create / elph_langevin_create has not been called"); return ELPH_E_STATE; }
    st->rng_on = true;
    st->rng_seed = seed;
    st->rng_batches = 0;
    return ELPH_OK;
}

extern "C" int elph_hmc_rng_batches(elph_handle h, uint64_t *batches) {
    CHECK_H(h);
    HmcState *st = state_of(h);
    if (!st || !batches) { elph_set_error("bad argument"); return ELPH_E_ARG; }
    *batches = st->rng_batches;
    return ELPH_OK;
}

extern "C" int elph_hmc_update_chains(elph_handle h, double dt, int64_t nt, int nb, double alpha, int use_precond, const double *R,
                                      const double *Rp, const double *Rm, const double *kpm_randn, const double *u_accept,
                                      int *accepted, double *iters_per_solve, double *energies, int *flag_out) {
    CHECK_H(h);
    HmcState *st = state_of(h);
    if (!st) { elph_set_error("elph_hmc_create has not been called"); return ELPH_E_STATE; }
    if (!st->have_state) { elph_set_error("elph_hmc_set_state(x) has not been called"); return ELPH_E_STATE; }
    if (!accepted || nt < 0 || nb < 1 || !(alpha >= 0.0 && alpha < 1.0) || !(dt > 0.0)) {
        elph_set_error("bad argument");
        return ELPH_E_ARG;
    }
    if ((!R || !Rp || !Rm || !u_accept || (use_precond && !kpm_randn)) && !st->rng_on) {
        elph_set_error("R, Rp, Rm, u_accept (and kpm_randn with a preconditioner) are required unless elph_hmc_set_rng was called");
        return ELPH_E_ARG;
    }
    if (use_precond && !h->kpm_created && !sharded(h)) { elph_set_error("elph_kpm_create has not been called"); return ELPH_E_STATE; }
    const int nch = st->nch;
    if (sharded(h)) {
        // one lattice over several ranks: the trajectory runs on the slab (own + ghost rows); the random vectors must be the slab's part
        // of the GLOBAL vectors (ghost entries included) and the uniform of the Metropolis test the same number on every rank
        const bool prec_ok = !use_precond || (elph_i_shard_full(h) && elph_i_shard_full(h)->kpm_created && kpm_randn && (!st->ssh || elph_i_shard_has_bonds(h)));
        if ((st->ssh && (!st->wown || st->shared)) || nch != 1 || !prec_ok || st->rng_on || !R || !Rp || !Rm || !u_accept) {
            elph_set_error("HMC on a sharded lattice: one chain, with R, Rp, Rm and u_accept given (the slab's part of the global vectors); bond "
                           "phonons after elph_shard_hmc_set_columns, without shared fields; a preconditioner needs elph_shard_set_full_lattice with "
                           "elph_kpm_create done, kpm_randn = start vectors of the WHOLE lattice and — bond phonons — elph_shard_set_bonds");
            return ELPH_E_UNSUPPORTED;
        }
    }
    RC(elph_i_ensure_capacity(h, 2 * nch));
    RC(elph_i_reserve_chains(h, nch));
    // nd: site vectors (ϕ±, R±, solutions); nfd: field vectors (x, v, dS/dx) of one chain
    const size_t nd = (size_t)h->ndim, nfd = (size_t)st->nf * (size_t)h->L, cbytes = nfd * sizeof(double), bytes = (size_t)nch * cbytes;
    const long long n = (long long)nfd * nch;
    const double dtp = dt / (double)nb;
    int64_t kpm_calls = 0;
    std::vector<int64_t> iters((size_t)nch, 0), itrs((size_t)nch, 0);
    std::vector<int> flag((size_t)nch, 0), dead((size_t)nch, 0);
    std::vector<double> H0((size_t)nch, 0.0), H1((size_t)nch, 0.0), S((size_t)nch, 0.0), K((size_t)nch, 0.0), P((size_t)nch, 0.0);
    auto bury = [&]() -> int {              // chains that just failed: remember the flag, restore their field
        for (int c = 0; c < nch; ++c)
            if (flag[(size_t)c] > 0 && !dead[(size_t)c]) {
                dead[(size_t)c] = flag[(size_t)c];
                HIPCHK(hipMemcpyAsync(st->x + (size_t)c * nfd, st->x0 + (size_t)c * nfd, cbytes, hipMemcpyDeviceToDevice, h->stream));
            }
        return ELPH_OK;
    };
    auto all_dead = [&]() { for (int c = 0; c < nch; ++c) if (!dead[(size_t)c]) return false; return true; };

    RC(update_model(h, st));
    // refresh_v!  (HMC.jl:648-659)
    RC(randn_vectors(h, st, st->y, R, nch, st->nf));
    RC(fa(h, st, st->y, st->y, -0.5));
    hipLaunchKernelGGL(k_hmc_refresh_v, dim3(nblk(n)), dim3(TPB), 0, h->stream, st->v, st->y, alpha, n);
    RC(chk("k_hmc_refresh_v"));
    HIPCHK(hipMemcpyAsync(st->x0, st->x, bytes, hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(st->v0, st->v, bytes, hipMemcpyDeviceToDevice, h->stream));
    // refresh_ϕ!  (HMC.jl:665-692): ϕ± = Λ⁻¹ Mᵀ R±      (R2, ϕ: [sign][chain][ndim])
    RC(randn_vectors(h, st, st->R2, Rp, nch));
    RC(randn_vectors(h, st, st->R2 + (size_t)nch * nd, Rm, nch));
    std::vector<double> kpm_own, u_own;      // generated in this order after R, R₊, R₋ when the caller left them NULL
    if (use_precond && !kpm_randn) {
        RC(randn_host(st, kpm_own, (size_t)(nt + 2) * 2 * (size_t)nch * (size_t)h->N));
        kpm_randn = kpm_own.data();
    }
    if (!u_accept) {
        RC(uniform_host(st, u_own, (size_t)nch));
        u_accept = u_own.data();
    }
    RC(elph_launch_mul(h, 1, h->d_b, st->R2, 2 * nch));
    if (st->ssh) {      // Λ⁻¹ ≡ 1: ϕ± = MᵀR±
        HIPCHK(hipMemcpyAsync(st->phi, h->d_b, 2 * (size_t)nch * nd * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    } else {
        hipLaunchKernelGGL(k_hmc_phi, dim3(nblk((long long)nd * nch), 2), dim3(TPB), 0, h->stream, st->phi, h->d_b, st->x, h->d_lam,
                           (int)h->N, (int)h->L, st->dtau, nch);
        RC(chk("k_hmc_phi"));
    }
    // a sharded lattice: MᵀR± is exact on the own rows only (its ghost rows would need R beyond the slab); ϕ± of the ghost rows from their owners
    if (sharded(h)) RC(elph_i_shard_ghost_sync(h, st->phi, 2));

    RC(calc_OinvLphi(h, st, use_precond, 2.0, kpm_randn, &kpm_calls, itrs.data(), flag.data()));
    // standard_update! :373 counts these iterations;  multitimestep_update! :507 has "iters += iters": not counted
    if (nb == 1) for (int c = 0; c < nch; ++c) iters[(size_t)c] = itrs[(size_t)c];
    RC(bury());
    if (!all_dead()) {
        RC(calc_H(h, st, H0.data(), S.data(), K.data()));
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
            for (int c = 0; c < nch; ++c)       // a dead chain idles at its old field
                if (dead[(size_t)c])
                    HIPCHK(hipMemcpyAsync(st->x + (size_t)c * nfd, st->x0 + (size_t)c * nfd, cbytes, hipMemcpyDeviceToDevice, h->stream));
            RC(update_model(h, st));
            RC(calc_OinvLphi(h, st, use_precond, 1.0, kpm_randn, &kpm_calls, itrs.data(), flag.data()));
            for (int c = 0; c < nch; ++c) if (!dead[(size_t)c]) iters[(size_t)c] += itrs[(size_t)c];
            RC(bury());
            if (all_dead()) break;                                             // :405-408
            RC(force(h, st, nb == 1));
            RC(leap(h, st, dt / 2, 0.0));                                      // :418
        }
    }
    if (!all_dead()) {
        RC(calc_OinvLphi(h, st, use_precond, 2.0, kpm_randn, &kpm_calls, itrs.data(), flag.data()));
        for (int c = 0; c < nch; ++c) if (!dead[(size_t)c]) iters[(size_t)c] += itrs[(size_t)c];
        RC(bury());
        if (!all_dead()) {
            RC(calc_H(h, st, H1.data(), S.data(), K.data()));
            for (int c = 0; c < nch; ++c) {
                if (dead[(size_t)c]) continue;
                const double e = exp(-(H1[(size_t)c] - H0[(size_t)c]));
                P[(size_t)c] = (1.0 < e) ? 1.0 : e;                            // min(1, exp(-ΔH)); NaN -> NaN -> reject, like Julia
            }
        }
    }
    bool any_reject = false;
    for (int c = 0; c < nch; ++c) {
        const int acc = (!dead[(size_t)c] && u_accept[c] < P[(size_t)c]) ? 1 : 0;      // :441
        accepted[c] = acc;
        if (!acc) {
            any_reject = true;
            HIPCHK(hipMemcpyAsync(st->x + (size_t)c * nfd, st->x0 + (size_t)c * nfd, cbytes, hipMemcpyDeviceToDevice, h->stream));
            hipLaunchKernelGGL(k_hmc_neg, dim3(nblk((long long)nfd)), dim3(TPB), 0, h->stream, st->v + (size_t)c * nfd,
                               st->v0 + (size_t)c * nfd, (long long)nfd);
            RC(chk("k_hmc_neg"));
        }
    }
    if (any_reject) RC(update_model(h, st));
    HIPCHK(hipStreamSynchronize(h->stream));
    for (int c = 0; c < nch; ++c) {
        if (iters_per_solve) iters_per_solve[c] = (double)((iters[(size_t)c] + (nt + 2) - 1) / (nt + 2));   // T1(cld(iters, Nt+2))
        if (energies) {
            double *e = energies + 5 * (size_t)c;
            e[0] = H0[(size_t)c]; e[1] = H1[(size_t)c]; e[2] = S[(size_t)c]; e[3] = K[(size_t)c]; e[4] = P[(size_t)c];
        }
        if (flag_out) flag_out[c] = dead[(size_t)c];
    }
    return ELPH_OK;
}

extern "C" int elph_hmc_update(elph_handle h, double dt, int64_t nt, int nb, double alpha, int use_precond, const double *R,
                               const double *Rp, const double *Rm, const double *kpm_randn, double u_accept, int *accepted,
                               double *iters_per_solve, double *energies, int *flag_out) {
    CHECK_H(h);
    HmcState *st = state_of(h);
    if (st && st->nch != 1) { elph_set_error("%d chains: use elph_hmc_update_chains", st->nch); return ELPH_E_STATE; }
    return elph_hmc_update_chains(h, dt, nt, nb, alpha, use_precond, R, Rp, Rm, kpm_randn, u_accept >= 0.0 ? &u_accept : nullptr, accepted, iters_per_solve,
                                  energies, flag_out);
}


// ==========================================================================================
// Langevin dynamics (LangevinDynamics.jl) on the device — Holstein.  Shares the HMC state (x, scratch vectors, ω, ω₄,
// the accelerator table, here FourierAccelerator.Q) and its helpers.
//   calc_dSdx! = calc_dSfdx! (:350-384: one solve MᵀM x = Mᵀg, dSf/dx = -2 gᵀ(∂M/∂x)M⁻¹g, the flag is ignored) + shifted
//                calc_dSbdx! (:334-344)
//   evolve!    = EulerDynamics (:81-130), RungeKuttaDynamics (:162-232), HeunsDynamics (:272-328)
// ==========================================================================================

namespace {

__global__ void __launch_bounds__(TPB) k_lv_axpby(double *__restrict__ out, double a, const double *__restrict__ x, double b,
                                                  const double *__restrict__ y, double c, const double *__restrict__ z, long long n) {
    const long long i = (long long)blockIdx.x * TPB + threadIdx.x;
    if (i < n) out[i] = a * x[i] + b * y[i] + (z ? c * z[i] : 0.0);
}

int axpby(elph_handle_s *h, double *out, double a, const double *x, double b, const double *y, double c, const double *z, long long n) {
    hipLaunchKernelGGL(k_lv_axpby, dim3(nblk(n)), dim3(TPB), 0, h->stream, out, a, x, b, y, c, z, n);
    return chk("k_lv_axpby");
}

// dS (layout S) = calc_dSdx!(x; g) for every chain (g: [nch][Ndim] on the host); per-chain iteration counts and flags
int langevin_force(elph_handle_s *h, HmcState *st, double *dS, const double *g_host, int use_precond, const double *bmax,
                   const double *bmin, int64_t *iters, int *flag) {
    const size_t nd = (size_t)h->ndim;
    const int nch = st->nch;
    RC(randn_vectors(h, st, st->R2, g_host, nch));                              // g in layout S
    if (use_precond) RC(elph_kpm_setup_chains(h, bmax, bmin, nullptr, nullptr, nullptr, nullptr, nullptr));   // setup!(P), :366
    RC(elph_launch_mul(h, 1, h->d_b, st->R2, nch));                             // Mᵀg (model.v″, :378)
    HIPCHK(hipMemsetAsync(h->d_x, 0, (size_t)nch * nd * sizeof(double), h->stream));   // fill!(M⁻¹g, 0), :367
    h->x_zero = true;
    std::vector<double> res((size_t)nch);
    RC(elph_i_ldiv_core(h, nch, use_precond ? 1 : 0, 0, iters, res.data(), flag));
    const long long n = (long long)st->nf * h->L * nch;
    if (st->ssh) {      // muldMdx!(dSfdx, g, ssh, M⁻¹g) (SSHModels.jl:707-829) with u = g given; no shifted term for bond phonons
        RC(elph_launch_force_ssh(h, h->d_p, h->d_x, st->R2, nch));
        RC(elph_launch_ssh_scatter(h, dS, h->d_p, st->x, h->d_ssh_par, h->d_ssh_cb, st->nf, st->dtau, 1, -2.0, nch));
        RC(alias_sum(h, st, dS));
        hipLaunchKernelGGL(k_hmc_dsb, dim3(nblk(n)), dim3(TPB), 0, h->stream, dS, st->x, st->par, st->nf, (int)h->L, st->dtau, 1, nch,
                           (const double *)nullptr);
    } else {
        RC(elph_launch_dmdx_holstein(h, dS, st->R2, h->d_x, st->x, st->dtau, -2.0, nch));   // -2 gᵀ(∂M/∂x)M⁻¹g, :381-384
        hipLaunchKernelGGL(k_hmc_dsb, dim3(nblk(n)), dim3(TPB), 0, h->stream, dS, st->x, st->par, (int)h->N, (int)h->L, st->dtau, 1, nch,
                           (const double *)h->d_lam);                           // calc_dSbdx!(dSdx, model, true), :341
    }
    RC(chk("k_hmc_dsb(langevin)"));
    return ELPH_OK;
}

}  // namespace

// LangevinDynamics on a Holstein handle: same arguments as elph_hmc_create, with fa_Q = FourierAccelerator.Q (evolve! calls
// fourier_accelerate! without use_mass).  elph_hmc_set_state / elph_hmc_get_state move the field x.
extern "C" int elph_langevin_create(elph_handle h, const double *omega, const double *omega4, const double *lambda,
                                    const double *lambda2, const double *mu, double dtau, const double *fa_Q) {
    return elph_hmc_create_chains(h, 1, omega, omega4, lambda, lambda2, mu, dtau, fa_Q);
}

// nchains independent Langevin trajectories of one deck in lockstep (every step one batched solve of nchains right-hand sides,
// one KPM expansion per chain); state and random vectors are chain-major: x, eta [nchains*Ndof], g1, g2 [nchains*Ndim],
// kpm_randn [2 set-ups][b_max|b_min][nchains][nsites]; iters, flag [nchains].
extern "C" int elph_langevin_create_chains(elph_handle h, int nchains, const double *omega, const double *omega4, const double *lambda,
                                           const double *lambda2, const double *mu, double dtau, const double *fa_Q) {
    return elph_hmc_create_chains(h, nchains, omega, omega4, lambda, lambda2, mu, dtau, fa_Q);
}

// The same for an SSH handle: the arguments of elph_hmc_create_ssh with fa_Q (per phonon) in place of the mass table.
extern "C" int elph_langevin_create_ssh(elph_handle h, int64_t nph, const double *omega, const double *omega4, const int64_t *cb_index,
                                        const double *t_ph, const double *alpha, const double *alpha2, const double *t_bare_cb,
                                        const double *mu, double dtau, const double *fa_Q) {
    return elph_hmc_create_ssh(h, nph, omega, omega4, cb_index, t_ph, alpha, alpha2, t_bare_cb, mu, dtau, fa_Q);
}

// evolve!(model, dyn, fa, P): scheme 0 Euler, 1 Runge-Kutta, 2 Heun.  The random numbers the reference draws are inputs:
// eta [Ndof] (randn!(η, model)), g1, g2 [Ndim] (the noise vectors of the first / second force estimate; g2 unused by Euler),
// kpm_randn [2][2][nsites] (b_max, b_min of the first and second setup!(P); NULL without preconditioner).
// iters: the count evolve! returns (Euler: the solve; RK: the second solve; Heun: div(it1 + it2, 2)); flag: last ldiv! flag
// (the reference ignores it: a failed solve contributes M⁻¹g = 0).
extern "C" int elph_langevin_evolve(elph_handle h, int scheme, double dt, int use_precond, const double *eta, const double *g1,
                                    const double *g2, const double *kpm_randn, int64_t *iters, int *flag) {
    CHECK_H(h);
    HmcState *st = state_of(h);
    if (!st) { elph_set_error("elph_langevin_create[_ssh|_chains] has not been called on this handle"); return ELPH_E_STATE; }
    if (!st->have_state) { elph_set_error("elph_hmc_set_state(x) has not been called"); return ELPH_E_STATE; }
    if (scheme < 0 || scheme > 2 || !(dt > 0.0)) { elph_set_error("bad argument"); return ELPH_E_ARG; }
    if ((!eta || !g1 || (scheme > 0 && !g2) || (use_precond && !kpm_randn)) && !st->rng_on) {
        elph_set_error("eta, g1 (g2 beyond Euler, kpm_randn with a preconditioner) are required unless elph_hmc_set_rng was called");
        return ELPH_E_ARG;
    }
    if (use_precond && !h->kpm_created) { elph_set_error("elph_kpm_create has not been called"); return ELPH_E_STATE; }
    const int nch = st->nch;
    RC(elph_i_ensure_capacity(h, std::max(2, 2 * nch)));
    RC(elph_i_reserve_chains(h, nch));
    const size_t N = (size_t)h->N;
    const long long n = (long long)st->nf * h->L * nch;      // field vectors of all chains (Holstein: nch * ndim)
    const double s2 = sqrt(2.0 * dt);
    std::vector<double> kpm_own;             // batches are drawn in the order eta, kpm_randn, g1, g2
    RC(update_model(h, st));
    RC(randn_vectors(h, st, st->v, eta, nch, st->nf));
    if (use_precond && !kpm_randn) {
        RC(randn_host(st, kpm_own, 4 * (size_t)nch * N));
        kpm_randn = kpm_own.data();
    }
    // start vectors of the two set-ups: [set-up][b_max | b_min][chain][N]
    const double *bm1 = kpm_randn, *bn1 = kpm_randn ? kpm_randn + (size_t)nch * N : nullptr;
    const double *bm2 = kpm_randn ? kpm_randn + 2 * (size_t)nch * N : nullptr, *bn2 = kpm_randn ? kpm_randn + 3 * (size_t)nch * N : nullptr;
    double *F1 = st->dS, *F2 = st->y, *xi = st->v, *dx = st->v0;      // the momentum buffers are free: Langevin has none
    std::vector<int64_t> it1((size_t)nch, 0), it2((size_t)nch, 0);
    std::vector<int> fl((size_t)nch, 0);
    if (scheme == 0) {
        RC(langevin_force(h, st, F1, g1, use_precond, bm1, bn1, it1.data(), fl.data()));
        RC(fa(h, st, F1, F1, 1.0));
        RC(fa(h, st, xi, xi, 0.5));
        RC(axpby(h, st->x, 1.0, st->x, s2, xi, -dt, F1, n));                       // x += √(2Δt) Q^½η − Δt Q dS/dx
    } else if (scheme == 1) {
        RC(langevin_force(h, st, F1, g1, use_precond, bm1, bn1, it1.data(), fl.data()));
        RC(axpby(h, dx, s2, xi, -dt, F1, 0.0, nullptr, n));                        // Δx = √(2Δt) η − Δt dS/dx   (no acceleration)
        RC(axpby(h, st->x, 1.0, st->x, 1.0, dx, 0.0, nullptr, n));
        RC(update_model(h, st));
        RC(langevin_force(h, st, F2, g2, use_precond, bm2, bn2, it2.data(), fl.data()));
        RC(axpby(h, st->x, 1.0, st->x, -1.0, dx, 0.0, nullptr, n));                // x = x′ − Δx
        RC(axpby(h, F1, 0.5, F2, 0.5, F1, 0.0, nullptr, n));                       // (dSdx′ + dSdx)/2
        RC(fa(h, st, F1, F1, 1.0));
        RC(fa(h, st, xi, xi, 0.5));
        RC(axpby(h, st->x, 1.0, st->x, s2, xi, -dt, F1, n));
        it1 = it2;
    } else {
        RC(fa(h, st, xi, xi, 0.5));                                                // ξ = Q^½η
        RC(langevin_force(h, st, F1, g1, use_precond, bm1, bn1, it1.data(), fl.data()));
        RC(fa(h, st, F1, F1, 1.0));                                                // dΓ/dx
        RC(axpby(h, dx, s2, xi, -dt, F1, 0.0, nullptr, n));
        RC(axpby(h, st->x, 1.0, st->x, 1.0, dx, 0.0, nullptr, n));
        RC(update_model(h, st));
        RC(langevin_force(h, st, F2, g2, use_precond, bm2, bn2, it2.data(), fl.data()));
        RC(fa(h, st, F2, F2, 1.0));
        RC(axpby(h, st->x, 1.0, st->x, -1.0, dx, 0.0, nullptr, n));                // x = x′ − Δx
        RC(axpby(h, F1, 0.5, F1, 0.5, F2, 0.0, nullptr, n));                       // (dΓ + dΓ′)/2
        RC(axpby(h, st->x, 1.0, st->x, s2, xi, -dt, F1, n));
        for (int c = 0; c < nch; ++c) it1[(size_t)c] = (it1[(size_t)c] + it2[(size_t)c]) / 2;
    }
    RC(update_model(h, st));
    HIPCHK(hipStreamSynchronize(h->stream));
    for (int c = 0; c < nch; ++c) {
        if (iters) iters[c] = it1[(size_t)c];
        if (flag) flag[c] = fl[(size_t)c];
    }
    return ELPH_OK;
}


// ==========================================================================================
// Special updates (SpecialUpdates.jl): one proposed move on the device-resident field of the HMC state
// ==========================================================================================

namespace {

// reflect column ci (kind 0) or swap columns ci, cj (kind 1) of a tau-major field vector x[t * nf + column]
__global__ void __launch_bounds__(TPB) k_hmc_colop(double *__restrict__ x, int nf, int L, int kind, int ci, int cj) {
    const int t = blockIdx.x * TPB + threadIdx.x;
    if (t >= L) return;
    double *row = x + (size_t)t * nf;
    if (kind == 0) row[ci] = -row[ci];
    else { const double a = row[ci]; row[ci] = row[cj]; row[cj] = a; }
}

}  // namespace

// The body of the loops of special_update! (SpecialUpdates.jl:103-136 reflection, :205-275 swap) for the single chain of an
// HMC state created with elph_hmc_create / elph_hmc_create_ssh:
//   S₀ = refresh_ϕ!(hmc, model, sample_R = true) — Rp, Rm are the fresh R± (HMC.jl:665-692), S₀ = (R₊² + R₋²)/2 + S_b;
//   the move (kind 0: x_col(τ) → −x_col(τ) on column col_i; kind 1: columns col_i, col_j exchange their world lines; 0-based
//   phonon columns = sites for Holstein), update_model!, calc_O⁻¹Λϕ!(…, 2.0), S₁ = calc_S;
//   accepted iff u < min(1, e^{−(S₁−S₀)}) and flag == 0, otherwise the move is undone and update_model! runs again.
// kpm_randn: b_max, b_min [2][nsites] of the one setup!(P) (NULL without preconditioner).  The choice of sites / bonds
// (sample!(model.rng, …)) stays with the caller.  out: accepted, S₀, S₁, iterations, flag.
extern "C" int elph_hmc_special_move_chains(elph_handle h, int kind, const int64_t *col_i, const int64_t *col_j, const double *Rp,
                                            const double *Rm, int use_precond, const double *kpm_randn, const double *u_accept,
                                            int *accepted, double *S0_out, double *S1_out, int64_t *iters_out, int *flag_out) {
    CHECK_H(h);
    HmcState *st = state_of(h);
    if (!st) { elph_set_error("elph_hmc_create[_ssh][_chains] has not been called"); return ELPH_E_STATE; }
    if (!st->have_state) { elph_set_error("elph_hmc_set_state(x) has not been called"); return ELPH_E_STATE; }
    const int nch = st->nch;
    if (!accepted || !col_i || kind < 0 || kind > 1 || (kind == 1 && !col_j)) { elph_set_error("bad argument"); return ELPH_E_ARG; }
    for (int c = 0; c < nch; ++c)
        if (col_i[c] < 0 || col_i[c] >= st->nf || (kind == 1 && (col_j[c] < 0 || col_j[c] >= st->nf))) {
            elph_set_error("chain %d: column outside 0..%d", c, st->nf - 1);
            return ELPH_E_ARG;
        }
    if (st->shared) {
        elph_set_error("special moves on shared fields: the reference's swap leaves the fields unequal and then stops in update_model!");
        return ELPH_E_UNSUPPORTED;
    }
    if ((!Rp || !Rm || !u_accept || (use_precond && !kpm_randn)) && !st->rng_on) {
        elph_set_error("Rp, Rm, u_accept (kpm_randn with a preconditioner) are required unless elph_hmc_set_rng was called");
        return ELPH_E_ARG;
    }
    if (use_precond && !h->kpm_created) { elph_set_error("elph_kpm_create has not been called"); return ELPH_E_STATE; }
    RC(elph_i_ensure_capacity(h, 2 * nch));
    RC(elph_i_reserve_chains(h, nch));
    const size_t nd = (size_t)h->ndim, nfd = (size_t)st->nf * (size_t)h->L;
    const int L = (int)h->L;
    RC(update_model(h, st));
    // refresh_ϕ!(…, sample_R = true) for every chain   (R2, ϕ: [sign][chain][ndim])
    RC(randn_vectors(h, st, st->R2, Rp, nch));
    RC(randn_vectors(h, st, st->R2 + (size_t)nch * nd, Rm, nch));
    std::vector<double> kpm_own, u_own;      // batches in the order Rp, Rm, kpm_randn, u_accept
    if (use_precond && !kpm_randn) {
        RC(randn_host(st, kpm_own, 2 * (size_t)nch * (size_t)h->N));
        kpm_randn = kpm_own.data();
    }
    if (!u_accept) {
        RC(uniform_host(st, u_own, (size_t)nch));
        u_accept = u_own.data();
    }
    RC(elph_launch_mul(h, 1, h->d_b, st->R2, 2 * nch));
    if (st->ssh) {
        HIPCHK(hipMemcpyAsync(st->phi, h->d_b, 2 * (size_t)nch * nd * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    } else {
        hipLaunchKernelGGL(k_hmc_phi, dim3(nblk((long long)nd * nch), 2), dim3(TPB), 0, h->stream, st->phi, h->d_b, st->x, h->d_lam, (int)h->N, L,
                           st->dtau, nch);
        RC(chk("k_hmc_phi"));
    }
    std::vector<double> rr((size_t)2 * nch), sf((size_t)2 * nch), sb0((size_t)nch), sb1((size_t)nch);
    RC(dots_host(h, st, st->R2, st->R2, (long long)nd, 2 * nch, rr.data()));
    RC(calc_Sb(h, st, sb0.data()));
    auto move = [&](const std::vector<int> &which) -> int {       // both moves are involutions: the same call undoes them
        for (int c = 0; c < nch; ++c) {
            if (!which[(size_t)c]) continue;
            hipLaunchKernelGGL(k_hmc_colop, dim3((unsigned)((L + TPB - 1) / TPB)), dim3(TPB), 0, h->stream, st->x + (size_t)c * nfd, st->nf, L,
                               kind, (int)col_i[c], (int)(col_j ? col_j[c] : 0));
            RC(chk("k_hmc_colop"));
        }
        return update_model(h, st);
    };
    RC(move(std::vector<int>((size_t)nch, 1)));
    int64_t kpm_calls = 0;
    std::vector<int64_t> iters((size_t)nch, 0);
    std::vector<int> flag((size_t)nch, 0), undo((size_t)nch, 0);
    RC(calc_OinvLphi(h, st, use_precond, 2.0, kpm_randn, &kpm_calls, iters.data(), flag.data()));
    RC(dots_host(h, st, h->d_b, h->d_x, (long long)nd, 2 * nch, sf.data()));
    RC(calc_Sb(h, st, sb1.data()));
    bool any_undo = false;
    for (int c = 0; c < nch; ++c) {
        const double S0 = rr[(size_t)c] / 2 + rr[(size_t)nch + c] / 2 + sb0[(size_t)c];
        const double S1 = sf[(size_t)c] / 2 + sf[(size_t)nch + c] / 2 + sb1[(size_t)c];
        const double e = exp(-(S1 - S0)), P = (1.0 < e) ? 1.0 : e;
        const int acc = (u_accept[c] < P && flag[(size_t)c] == 0) ? 1 : 0;
        accepted[c] = acc;
        undo[(size_t)c] = !acc;
        any_undo = any_undo || !acc;
        if (S0_out) S0_out[c] = S0;
        if (S1_out) S1_out[c] = S1;
        if (iters_out) iters_out[c] = iters[(size_t)c];
        if (flag_out) flag_out[c] = flag[(size_t)c];
    }
    if (any_undo) RC(move(undo));
    HIPCHK(hipStreamSynchronize(h->stream));
    return ELPH_OK;
}

extern "C" int elph_hmc_special_move(elph_handle h, int kind, int64_t col_i, int64_t col_j, const double *Rp, const double *Rm,
                                     int use_precond, const double *kpm_randn, double u_accept, int *accepted, double *S0_out,
                                     double *S1_out, int64_t *iters_out, int *flag_out) {
    CHECK_H(h);
    HmcState *st = state_of(h);
    if (!st || st->nch != 1) { elph_set_error("elph_hmc_create / elph_hmc_create_ssh (single chain) has not been called"); return ELPH_E_STATE; }
    return elph_hmc_special_move_chains(h, kind, &col_i, &col_j, Rp, Rm, use_precond, kpm_randn, u_accept >= 0.0 ? &u_accept : nullptr,
                                        accepted, S0_out, S1_out, iters_out, flag_out);
}
