// dft.hip — tau-axis transforms for the KPM preconditioner and Fourier acceleration.
//
// Reference: TimeFreqFFTs.jl:55-73 (tau_to_omega!: FFT of Theta.*v), :112-130 (omega_to_tau!),
// FourierAcceleration.jl:91-143 (fourier_accelerate!).  FFTW conventions: forward unnormalised
// exp(-2 pi i k t/L), inverse scaled 1/L.
//
// L_tau is 40..160 on every deck and all site columns are independent, so each transform is a direct
// real DFT over the half spectrum (real input => Hermitian symmetry; the twisted spectrum obeys
// nu[L-1-k] = conj nu[k], exactly the half the KPM loop visits, KPMPreconditioners.jl:449-467):
//   * lane = site: every global access is a coalesced 512-B (f64) / 1-KB (complex) row of layout S;
//   * twiddles come from host-built tables with the reduction index contiguous, Tk[k][t] / Tt[t][k],
//     so a wave reads them with wide *scalar* loads (the address depends only on block and loop index);
//     the inverse tables carry the Hermitian weight and the 1/L, so the inner loop is 2 FMA per term;
//   * the reduction axis is walked in register chunks with the next chunk's loads issued before the
//     current chunk's FMAs (two named buffers), so a wave pays ~L/TC memory round trips, not L.
// O(L^2) per column is deliberate at these lengths (13 MFLOP at config C); tables are O(L^2) bytes.

#include <cstdlib>

#include "elph_internal.h"

#define WAVE ELPH_WAVE

__device__ __forceinline__ double dft_wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}

// `state` (optional) points at the CURRENT copy of the CG state (the host adds the launch parity)
__device__ __forceinline__ bool dft_done(const CgState *state, int rhs) {
    if (!state) return false;
    return __hip_atomic_load(&state[2 * rhs].done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
}

// out[k][s] = f(k,s) * sum_t Tk[k][t] * v[t][s]      k in [0,K), real v
//   PLAIN: f = symmetrised diag^power (fourier_accelerate!), else f = 1
template <int KPT, int TC, bool PLAIN>
__global__ void __launch_bounds__(WAVE) k_dft_fwd_tab(double2 *__restrict__ out, const double *__restrict__ v,
                                                      const double2 *__restrict__ Tk, int N, int L, int K, int Lp,
                                                      const CgState *state, const double *__restrict__ diag, double power) {
    // Tk rows are zero-padded to Lp = roundup(L, 2*TC): the tail needs no predicate (clamped loads * 0)
    const int rhs = blockIdx.z;
    if (dft_done(state, rhs)) return;
    const int s = blockIdx.x * WAVE + threadIdx.x;
    const bool ok = s < N;
    const int sc = ok ? s : N - 1;
    const int k0 = blockIdx.y * KPT;
    const double *vv = v + (size_t)rhs * N * L;
    const double2 *tw[KPT];
    double2 acc[KPT];
#pragma unroll
    for (int kk = 0; kk < KPT; ++kk) {
        const int k = (k0 + kk < K) ? k0 + kk : K - 1;
        tw[kk] = Tk + (size_t)k * Lp;
        acc[kk] = make_double2(0.0, 0.0);
    }
    double xa[TC], xb[TC];
    auto load = [&](double (&x)[TC], int t0) {
#pragma unroll
        for (int j = 0; j < TC; ++j) {
            const int t = (t0 + j < L) ? t0 + j : L - 1;
            x[j] = vv[(size_t)t * N + sc];
        }
    };
    auto comp = [&](const double (&x)[TC], int t0) {
#pragma unroll
        for (int kk = 0; kk < KPT; ++kk) {
#pragma unroll
            for (int j = 0; j < TC; ++j) {
                const double2 w = tw[kk][t0 + j];
                acc[kk].x += x[j] * w.x;
                acc[kk].y += x[j] * w.y;
            }
        }
    };
    load(xa, 0);
    for (int t0 = 0; t0 < Lp; t0 += 2 * TC) {
        load(xb, t0 + TC);
        comp(xa, t0);
        load(xa, t0 + 2 * TC);
        comp(xb, t0 + TC);
    }
    if (ok) {
#pragma unroll
        for (int kk = 0; kk < KPT; ++kk) {
            const int k = k0 + kk;
            if (k < K) {
                double2 r = acc[kk];
                if (PLAIN && diag) {
                    // Re iFFT(D .* FFT(v)) of a real v only sees the symmetric part of D: (D[k] + D[L-k])/2
                    const int km = (k == 0) ? 0 : L - k;
                    const double f = 0.5 * (pow(diag[(size_t)k * N + s], power) + pow(diag[(size_t)km * N + s], power));
                    r.x *= f;
                    r.y *= f;
                }
                out[((size_t)rhs * K + k) * N + s] = r;
            }
        }
    }
}

// out[t][s] = sum_{k<K} Re( Tt[t][k] * nu[k][s] )   (weights and 1/L folded into Tt); optional fused partial r.out
template <int TPT, int KC>
__global__ void __launch_bounds__(WAVE) k_dft_inv_tab(double *__restrict__ out, const double2 *__restrict__ nu,
                                                      const double2 *__restrict__ Tt, int N, int L, int K, int Kp,
                                                      const CgState *state, const double *__restrict__ rvec,
                                                      double *__restrict__ rz_part, int nrz) {
    // Tt rows are zero-padded to Kp = roundup(K, 2*KC)
    const int rhs = blockIdx.z;
    if (dft_done(state, rhs)) return;
    const int s = blockIdx.x * WAVE + threadIdx.x;
    const bool ok = s < N;
    const int sc = ok ? s : N - 1;
    const int t0 = blockIdx.y * TPT;
    const double2 *nn = nu + (size_t)rhs * K * N;
    const double2 *tw[TPT];
    double acc[TPT];
#pragma unroll
    for (int tt = 0; tt < TPT; ++tt) {
        const int t = (t0 + tt < L) ? t0 + tt : L - 1;
        tw[tt] = Tt + (size_t)t * Kp;
        acc[tt] = 0.0;
    }
    double2 xa[KC], xb[KC];
    auto load = [&](double2 (&x)[KC], int kb) {
#pragma unroll
        for (int j = 0; j < KC; ++j) {
            const int k = (kb + j < K) ? kb + j : K - 1;
            x[j] = nn[(size_t)k * N + sc];
        }
    };
    auto comp = [&](const double2 (&x)[KC], int kb) {
#pragma unroll
        for (int tt = 0; tt < TPT; ++tt) {
#pragma unroll
            for (int j = 0; j < KC; ++j) {
                const double2 w = tw[tt][kb + j];
                acc[tt] += w.x * x[j].x - w.y * x[j].y;
            }
        }
    };
    load(xa, 0);
    for (int kb = 0; kb < Kp; kb += 2 * KC) {
        load(xb, kb + KC);
        comp(xa, kb);
        load(xa, kb + 2 * KC);
        comp(xb, kb + KC);
    }
    double dot = 0.0;
#pragma unroll
    for (int tt = 0; tt < TPT; ++tt) {
        const int t = t0 + tt;
        if (ok && t < L) {
            const size_t i = (size_t)rhs * N * L + (size_t)t * N + s;
            out[i] = acc[tt];
            if (rz_part) dot += rvec[i] * acc[tt];
        }
    }
    if (rz_part) {
        dot = dft_wave_sum(dot);
        if (threadIdx.x == 0) {
            // own slot, plus this block's share of the slots no block of this grid owns (the reducers read all nrz)
            const int G = (int)(gridDim.x * gridDim.y), b = (int)(blockIdx.y * gridDim.x + blockIdx.x);
            double *slots = rz_part + (size_t)rhs * nrz;
            slots[b] = dot;
            for (int q = G + b; q < nrz; q += G) slots[q] = 0.0;
        }
    }
}

static int dft_check(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        elph_set_error("launch %s failed: %s", what, hipGetErrorString(e));
        return ELPH_E_HIP;
    }
    return ELPH_OK;
}

// Single-solve shapes (2 outputs per wave: many waves for the latency-bound case).  These kernels are bound by the
// scalar twiddle stream (106 SGPRs hold < 30 twiddles); a wider shape (8 outputs per wave) was measured ~15 % SLOWER in
// a batch, so batches go to the matrix-core GEMM form in dft_mfma.hip instead.
constexpr int DFT_KPT = 2, DFT_TC = 40, DFT_TPT = 2, DFT_KC = 20;
static int dft_pad(int n, int m) { return ((n + m - 1) / m) * m; }
// nu[rhs][k][s] (half spectrum, k < ceil(L/2)) = FFT_t(Theta .* v)[k]
int elph_dft_fwd_twisted(elph_handle_s *h, double2 *nu, const double *vS, int N, int nrhs, const CgState *st) {
    const int L = (int)h->L, Lo2 = (L + 1) / 2, nst = (N + WAVE - 1) / WAVE;
    if (elph_dft_big(h)) return elph_dft_big_fwd(h, true, nu, vS, N, nrhs);       // (finished right-hand sides are transformed too)
    if (elph_dft_mfma_usable(h, 0, false, N, nrhs)) return elph_dft_mfma_fwd(h, 0, nu, vS, N, nrhs, st);
    if (elph_dft_mfma1_usable(h, false, N, 0)) return elph_dft_mfma1_fwd(h, nu, vS, N, nrhs, st);
    hipLaunchKernelGGL((k_dft_fwd_tab<DFT_KPT, DFT_TC, false>),
                           dim3((unsigned)nst, (unsigned)((Lo2 + DFT_KPT - 1) / DFT_KPT), (unsigned)nrhs), dim3(WAVE), 0, h->stream, nu,
                           vS, h->d_Tk, N, L, Lo2, dft_pad(L, 2 * DFT_TC), st, (const double *)nullptr, 0.0);
    return dft_check("k_dft_fwd_tab(twisted)");
}

// out = Re( conj(Theta) .* iFFT(nu) ) from the half spectrum; rz_part (optional) receives partial r.out sums
int elph_dft_inv_twisted(elph_handle_s *h, double *outS, const double2 *nu, int N, int nrhs, const CgState *st,
                         const double *rvec, double *rz_part, int nrz) {
    const int L = (int)h->L, Lo2 = (L + 1) / 2, nst = (N + WAVE - 1) / WAVE;
    if (elph_dft_big(h)) return elph_dft_big_inv(h, true, outS, nu, N, nrhs, rvec, rz_part, nrz);
    if (elph_dft_mfma_usable(h, 0, true, N, nrhs)) return elph_dft_mfma_inv(h, 0, outS, nu, N, nrhs, st, rvec, rz_part, nrz);
    if (elph_dft_mfma1_usable(h, true, N, rz_part ? nrz : 0)) return elph_dft_mfma1_inv(h, outS, nu, N, nrhs, st, rvec, rz_part, nrz);
    hipLaunchKernelGGL((k_dft_inv_tab<DFT_TPT, DFT_KC>),
                           dim3((unsigned)nst, (unsigned)((L + DFT_TPT - 1) / DFT_TPT), (unsigned)nrhs), dim3(WAVE), 0, h->stream, outS,
                           nu, h->d_Tt, N, L, Lo2, dft_pad(Lo2, 2 * DFT_KC), st, rvec, rz_part, nrz);
    return dft_check("k_dft_inv_tab(twisted)");
}

// u[rhs][k][s] *= (D[k][s]^p + D[L-k][s]^p)/2  — the symmetrised diagonal of fourier_accelerate! between the two GEMM-form
// transforms of a batch (the scalar forward kernel applies it in its epilogue)
__global__ void __launch_bounds__(256) k_dft_diag(double2 *__restrict__ u, const double *__restrict__ diag, double power, int N,
                                                  int L, int K, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int s = (int)(i % N), k = (int)((i / N) % K);
    const int km = (k == 0) ? 0 : L - k;
    const double f = 0.5 * (pow(diag[(size_t)k * N + s], power) + pow(diag[(size_t)km * N + s], power));
    double2 v = u[i];
    v.x *= f; v.y *= f;
    u[i] = v;
}

// out = Re iFFT( diag^power .* FFT(in) ), N columns, nvec vectors sharing one diagonal
int elph_dft_accel(elph_handle_s *h, double *outS, const double *inS, const double *diagS, double power, int N, double2 *u, int nvec) {
    const int L = (int)h->L, Lh = L / 2 + 1, nst = (N + WAVE - 1) / WAVE;
    if (elph_dft_big(h)) {
        int rc = elph_dft_big_fwd(h, false, u, inS, N, nvec);
        if (rc) return rc;
        const long long total = (long long)nvec * Lh * N;
        hipLaunchKernelGGL(k_dft_diag, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, h->stream, u, diagS, power, N, L, Lh, total);
        rc = dft_check("k_dft_diag");
        if (rc) return rc;
        return elph_dft_big_inv(h, false, outS, u, N, nvec, nullptr, nullptr, 0);
    }
    if (elph_dft_mfma_usable(h, 1, false, N, nvec) && elph_dft_mfma_usable(h, 1, true, N, nvec)) {
        int rc = elph_dft_mfma_fwd(h, 1, u, inS, N, nvec, nullptr);
        if (rc) return rc;
        const long long total = (long long)nvec * Lh * N;
        hipLaunchKernelGGL(k_dft_diag, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, h->stream, u, diagS, power, N, L, Lh, total);
        rc = dft_check("k_dft_diag");
        if (rc) return rc;
        return elph_dft_mfma_inv(h, 1, outS, u, N, nvec, nullptr, nullptr, nullptr, 0);
    }
    hipLaunchKernelGGL((k_dft_fwd_tab<DFT_KPT, DFT_TC, true>), dim3((unsigned)nst, (unsigned)((Lh + DFT_KPT - 1) / DFT_KPT), (unsigned)nvec),
                       dim3(WAVE), 0, h->stream, u, inS, h->d_Pk, N, L, Lh, dft_pad(L, 2 * DFT_TC), (const CgState *)nullptr, diagS, power);
    hipLaunchKernelGGL((k_dft_inv_tab<DFT_TPT, DFT_KC>), dim3((unsigned)nst, (unsigned)((L + DFT_TPT - 1) / DFT_TPT), (unsigned)nvec),
                       dim3(WAVE), 0, h->stream, outS, u, h->d_Pt, N, L, Lh, dft_pad(Lh, 2 * DFT_KC), (const CgState *)nullptr,
                       (const double *)nullptr, (double *)nullptr, 0);
    return dft_check("fourier_accelerate");
}

// nu[rhs][k][s] (half spectrum, k <= L/2) = FFT_t(v)[k]  — plain (untwisted) transform, no diagonal
int elph_dft_fwd_plain(elph_handle_s *h, double2 *nu, const double *vS, int N, int nrhs) {
    const int L = (int)h->L, Lh = L / 2 + 1, nst = (N + WAVE - 1) / WAVE;
    if (elph_dft_big(h)) return elph_dft_big_fwd(h, false, nu, vS, N, nrhs);
    if (elph_dft_mfma_usable(h, 1, false, N, nrhs)) return elph_dft_mfma_fwd(h, 1, nu, vS, N, nrhs, nullptr);
    hipLaunchKernelGGL((k_dft_fwd_tab<DFT_KPT, DFT_TC, true>),
                           dim3((unsigned)nst, (unsigned)((Lh + DFT_KPT - 1) / DFT_KPT), (unsigned)nrhs), dim3(WAVE), 0, h->stream, nu, vS,
                           h->d_Pk, N, L, Lh, dft_pad(L, 2 * DFT_TC), (const CgState *)nullptr, (const double *)nullptr, 0.0);
    return dft_check("k_dft_fwd_tab(plain)");
}

// out = Re iFFT(nu) from the half spectrum k <= L/2 (Hermitian weights and 1/L in the table)
int elph_dft_inv_plain(elph_handle_s *h, double *outS, const double2 *nu, int N, int nrhs) {
    const int L = (int)h->L, Lh = L / 2 + 1, nst = (N + WAVE - 1) / WAVE;
    if (elph_dft_big(h)) return elph_dft_big_inv(h, false, outS, nu, N, nrhs, nullptr, nullptr, 0);
    if (elph_dft_mfma_usable(h, 1, true, N, nrhs)) return elph_dft_mfma_inv(h, 1, outS, nu, N, nrhs, nullptr, nullptr, nullptr, 0);
    hipLaunchKernelGGL((k_dft_inv_tab<DFT_TPT, DFT_KC>),
                           dim3((unsigned)nst, (unsigned)((L + DFT_TPT - 1) / DFT_TPT), (unsigned)nrhs), dim3(WAVE), 0, h->stream, outS,
                           nu, h->d_Pt, N, L, Lh, dft_pad(Lh, 2 * DFT_KC), (const CgState *)nullptr, (const double *)nullptr,
                           (double *)nullptr, 0);
    return dft_check("k_dft_inv_tab(plain)");
}

// host: build the four twiddle tables with exact index reduction
int elph_dft_build_tables(elph_handle_s *h) {
    {   // long axes: one Cooley-Tukey split instead of O(L^2) tables (dft_big.hip) — always beyond 1024 slices; from 401 (where the matrix-core
        // forms end) to 1024 when the length has a divisor >= 4 below its square root: measured (round 6, profiles/r06/long_time_axes_split_from_401.log,
        // 16 x 16 sites, KPM apply of 16 right-hand sides): 480 slices 298 -> 219 us, 512: 350 -> 238, 800: 815 -> 394, 1000: 1328 -> 512 against the
        // scalar-twiddle kernels.  ELPH_DFT_BIG_FROM=n: the split for every length beyond n instead (A/B)
        const char *e = getenv("ELPH_DFT_BIG_FROM");
        bool big = h->L > 1024;
        if (e) big = h->L > atoll(e);
        else if (h->L > 400 && !big) {
            long long best = 0;
            for (long long f = 2; f * f <= h->L; ++f) if (h->L % f == 0) best = f;
            big = best >= 4;
        }
        if (big) return elph_dft_big_build_tables(h);
    }
    const int L = (int)h->L, Lo2 = (L + 1) / 2, Lh = L / 2 + 1;
    const int Lp = dft_pad(L, 2 * DFT_TC), Kp2 = dft_pad(Lo2, 2 * DFT_KC), Kph = dft_pad(Lh, 2 * DFT_KC);
    const double2 zero = make_double2(0.0, 0.0);
    std::vector<double2> Tk((size_t)Lo2 * Lp, zero), Tt((size_t)L * Kp2, zero), Pk((size_t)Lh * Lp, zero), Pt((size_t)L * Kph, zero);
    const double invL = 1.0 / (double)L;
    for (int k = 0; k < Lo2; ++k) {
        // odd L: the middle frequency k = (L-1)/2 is its own mirror image (weight 1)
        const double wgt = ((L & 1) && k == Lo2 - 1) ? 1.0 : 2.0;
        for (int t = 0; t < L; ++t) {
            const long long m = ((long long)(2 * k + 1) * t) % (2LL * L);
            const double a = M_PI * (double)m / (double)L;
            Tk[(size_t)k * Lp + t] = make_double2(cos(a), -sin(a));                            // exp(-i a)
            Tt[(size_t)t * Kp2 + k] = make_double2(wgt * invL * cos(a), wgt * invL * sin(a));  // wgt/L exp(+i a)
        }
    }
    for (int k = 0; k < Lh; ++k) {
        const double wgt = (k == 0 || 2 * k == L) ? 1.0 : 2.0;
        for (int t = 0; t < L; ++t) {
            const long long m = ((long long)k * t) % L;
            const double a = 2.0 * M_PI * (double)m / (double)L;
            Pk[(size_t)k * Lp + t] = make_double2(cos(a), -sin(a));
            Pt[(size_t)t * Kph + k] = make_double2(wgt * invL * cos(a), wgt * invL * sin(a));
        }
    }
    struct { double2 **d; std::vector<double2> *v; } tabs[] = {{&h->d_Tk, &Tk}, {&h->d_Tt, &Tt}, {&h->d_Pk, &Pk}, {&h->d_Pt, &Pt}};
    for (auto &tb : tabs) {
        if (*tb.d) { HIPCHK(hipFree(*tb.d)); *tb.d = nullptr; }
        HIPCHK(hipMalloc((void **)tb.d, tb.v->size() * sizeof(double2)));
        HIPCHK(hipMemcpy(*tb.d, tb.v->data(), tb.v->size() * sizeof(double2), hipMemcpyHostToDevice));
    }
    return elph_dft_mfma_build_tables(h);
}
