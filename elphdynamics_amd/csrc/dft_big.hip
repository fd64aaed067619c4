// dft_big.hip — tau-axis transforms for LONG time axes (L_tau > 1024), where the direct DFT of dft.hip / dft_mfma.hip would
// need O(L^2) twiddle tables and O(L^2) work per column.
//
// Reference: TimeFreqFFTs.jl:55-73,112-130 (tau_to_omega! / omega_to_tau!: FFTW plans take any length, :32-45) and
// FourierAcceleration.jl:91-143.  Same conventions as dft.hip: forward unnormalised exp(-2 pi i k t / L), inverse scaled 1/L,
// twisted transform = FFT of Theta .* v with Theta_t = exp(-i pi t / L).
//
// One Cooley-Tukey split L = L1 * L2 (both factors <= 1024, chosen near sqrt(L) on the host; a length without such a split —
// a prime — runs the direct transform with one table of L roots of unity, k_big_direct: any length works, those slowly):
//   t = L2 a + b,  k = c + L1 d:
//   X[c + L1 d] = sum_b  W2[d][b] * ( TW[(c b) mod L] * sum_a W1[c][a] u[L2 a + b] )
// i.e. L2 transforms of length L1 (stride L2), a twiddle, L1 transforms of length L2 — each a direct DFT with a small table
// (L1^2 + L2^2 + L entries instead of L^2), L (L1 + L2) complex multiply-adds per column instead of L^2.  The lane is the site, as
// everywhere in the transforms: every access is a coalesced row of layout S; twiddles are wave-uniform scalar loads.
// This is the fallback for axes the matrix-core forms do not reach — correctness first (the deck sizes of the reference are
// L_tau <= ~200); it runs full complex transforms and takes the half spectrum afterwards.

#include <cmath>
#include <cstdlib>
#include <vector>

#include "elph_internal.h"

#define WAVE ELPH_WAVE
#define RC(call)                \
    do {                        \
        int _rc = (call);       \
        if (_rc) return _rc;    \
    } while (0)

namespace {

int big_check(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { elph_set_error("launch %s failed: %s", what, hipGetErrorString(e)); return ELPH_E_HIP; }
    return ELPH_OK;
}

// u[t][s] = (TWISTED ? Theta_t : 1) * v[t][s]
template <bool TWISTED>
__global__ void __launch_bounds__(256) k_big_load(double2 *__restrict__ u, const double *__restrict__ v, const double2 *__restrict__ theta,
                                                  int N, int L, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int t = (int)((i / N) % L);
    const double x = v[i];
    if (TWISTED) { const double2 th = theta[t]; u[i] = make_double2(th.x * x, th.y * x); }
    else u[i] = make_double2(x, 0.0);
}

// step 1: Z[c][b] = TW[(c b) mod L]^(+-1) * sum_a W1[c][a]^(+-1) u[L2 a + b]      (row index of Z: c * L2 + b)
// grid: x = site tile, y = output row (c, b), z = vector; one wave per block
template <bool INV>
__global__ void __launch_bounds__(WAVE) k_big_step1(double2 *__restrict__ Z, const double2 *__restrict__ u, const double2 *__restrict__ W1,
                                                    const double2 *__restrict__ TW, int N, int L, int L1, int L2) {
    const int s = blockIdx.x * WAVE + threadIdx.x;
    const int row = blockIdx.y, c = row / L2, b = row - c * L2;
    const size_t base = (size_t)blockIdx.z * L * N;
    const int sc = (s < N) ? s : N - 1;
    const double2 *w = W1 + (size_t)c * L1;
    double ax = 0.0, ay = 0.0;
    for (int a = 0; a < L1; ++a) {
        const double2 x = u[base + (size_t)(L2 * a + b) * N + sc];
        const double2 ww = w[a];
        const double wy = INV ? -ww.y : ww.y;
        ax += ww.x * x.x - wy * x.y;
        ay += ww.x * x.y + wy * x.x;
    }
    const double2 tw = TW[(int)(((long long)c * b) % L)];
    const double ty = INV ? -tw.y : tw.y;
    if (s < N) Z[base + (size_t)row * N + s] = make_double2(tw.x * ax - ty * ay, tw.x * ay + ty * ax);
}

// step 2: X[c + L1 d] = scale * sum_b W2[d][b]^(+-1) Z[c][b]
template <bool INV>
__global__ void __launch_bounds__(WAVE) k_big_step2(double2 *__restrict__ X, const double2 *__restrict__ Z, const double2 *__restrict__ W2,
                                                    int N, int L, int L1, int L2, double scale) {
    const int s = blockIdx.x * WAVE + threadIdx.x;
    const int k = blockIdx.y, d = k / L1, c = k - d * L1;
    const size_t base = (size_t)blockIdx.z * L * N;
    const int sc = (s < N) ? s : N - 1;
    const double2 *w = W2 + (size_t)d * L2;
    double ax = 0.0, ay = 0.0;
    for (int b = 0; b < L2; ++b) {
        const double2 x = Z[base + (size_t)(c * L2 + b) * N + sc];
        const double2 ww = w[b];
        const double wy = INV ? -ww.y : ww.y;
        ax += ww.x * x.x - wy * x.y;
        ay += ww.x * x.y + wy * x.x;
    }
    if (s < N) X[base + (size_t)k * N + s] = make_double2(scale * ax, scale * ay);
}

// lengths without such a split (a prime L > 1024): the direct transform with the one table of L roots of unity,
// X[k] = scale * sum_t TW[(k t) mod L]^(+-1) u[t] — O(L^2) per column, O(L) table; correct for any length, slow
template <bool INV>
__global__ void __launch_bounds__(WAVE) k_big_direct(double2 *__restrict__ X, const double2 *__restrict__ u, const double2 *__restrict__ TW,
                                                     int N, int L, double scale) {
    const int s = blockIdx.x * WAVE + threadIdx.x;
    const int k = blockIdx.y;
    const size_t base = (size_t)blockIdx.z * L * N;
    const int sc = (s < N) ? s : N - 1;
    double ax = 0.0, ay = 0.0;
    int idx = 0;
    for (int t = 0; t < L; ++t) {
        const double2 x = u[base + (size_t)t * N + sc];
        const double2 ww = TW[idx];
        const double wy = INV ? -ww.y : ww.y;
        ax += ww.x * x.x - wy * x.y;
        ay += ww.x * x.y + wy * x.x;
        idx += k;
        if (idx >= L) idx -= L;
    }
    if (s < N) X[base + (size_t)k * N + s] = make_double2(scale * ax, scale * ay);
}

// half spectrum out of the full one: nu[vec][k][s] = X[vec][k][s], k < K
__global__ void __launch_bounds__(256) k_big_take(double2 *__restrict__ nu, const double2 *__restrict__ X, int N, int L, int K, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int s = (int)(i % N), k = (int)((i / N) % K);
    const long long vec = i / ((long long)N * K);
    nu[i] = X[((size_t)vec * L + k) * N + s];
}

// full spectrum from the half one.  TWISTED: nu[L-1-k] = conj nu[k] (k < K = ceil(L/2));  plain: nu[L-k] = conj nu[k] (k <= L/2)
template <bool TWISTED>
__global__ void __launch_bounds__(256) k_big_expand(double2 *__restrict__ X, const double2 *__restrict__ nu, int N, int L, int K, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int s = (int)(i % N), k = (int)((i / N) % L);
    const long long vec = i / ((long long)N * L);
    const double2 *h = nu + (size_t)vec * K * N;
    double2 v;
    if (k < K) v = h[(size_t)k * N + s];
    else {
        const int km = TWISTED ? L - 1 - k : L - k;
        const double2 c = h[(size_t)km * N + s];
        v = make_double2(c.x, -c.y);
    }
    X[i] = v;
}

// out[t][s] = Re( (TWISTED ? conj(Theta_t) : 1) * y[t][s] ); optional partial r.out per time slice (slot t; slots >= L zeroed)
// grid: x = time slice, y = vector; one wave per block walking the sites
template <bool TWISTED>
__global__ void __launch_bounds__(WAVE) k_big_store(double *__restrict__ out, const double2 *__restrict__ y, const double2 *__restrict__ theta,
                                                    int N, int L, const double *__restrict__ rvec, double *__restrict__ rz_part, int nrz) {
    const int t = blockIdx.x;
    const size_t base = ((size_t)blockIdx.y * L + t) * N;
    const double2 th = TWISTED ? theta[t] : make_double2(1.0, 0.0);
    double dot = 0.0;
    for (int s = threadIdx.x; s < N; s += WAVE) {
        const double2 v = y[base + s];
        const double o = th.x * v.x + th.y * v.y;          // Re(conj(theta) v)
        out[base + s] = o;
        if (rz_part) dot += rvec[base + s] * o;
    }
    if (rz_part) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) dot += __shfl_xor(dot, o, WAVE);
        if (threadIdx.x == 0) {
            double *slots = rz_part + (size_t)blockIdx.y * nrz;
            slots[t] = dot;
            for (int q = L + t; q < nrz; q += L) slots[q] = 0.0;
        }
    }
}

// ---- round 6: the two steps register-blocked and fused with their neighbours ---------------------------------------------------------------
// The kernels above produce ONE output row per wave: every input row of a sub-transform is read L1 (L2) times, and the transform is four
// kernels with full complex intermediates (load | step 1 | step 2 | take).  Here a wave produces BT output rows of its sub-transform at once —
// the input row is loaded once per BT outputs and meets BT wave-uniform twiddles (32 fma per 16-byte load at BT = 8: arithmetic, not
// re-reads, bounds the step) — and the neighbours ride along: step 1 takes its input straight from the caller's array (forward: the real
// vector times Theta; inverse: the half spectrum with its mirror image), step 2 writes the caller's array (forward: the half spectrum only —
// rows beyond it are not computed; inverse: Re(conj(Theta) y) and the r.out partials).  Two kernels and one complex intermediate per
// transform.  Per output element the sums run in the same order with the same expressions as above.
constexpr int BT = 8;

// step 1:  Z[c][b] = TW[(c b) mod L]^(+-1) sum_a W1[c][a]^(+-1) u[L2 a + b]   for c = c0 .. c0 + BT - 1
//   forward: u[t] = (TWISTED ? Theta_t : 1) v[t] (v real);  inverse: u[k] = the full spectrum out of the half one (nu[k], k < K; else conj of its mirror)
// grid: x = site tile, y = (c group) * L2 + b, z = vector
template <bool INV, bool TWISTED>
__global__ void __launch_bounds__(WAVE) k_big_s1(double2 *__restrict__ Z, const double *__restrict__ vreal, const double2 *__restrict__ nu,
                                                 const double2 *__restrict__ W1, const double2 *__restrict__ TW, const double2 *__restrict__ theta,
                                                 int N, int L, int L1, int L2, int K) {
    const int s = blockIdx.x * WAVE + threadIdx.x;
    const int cg = blockIdx.y / L2, b = blockIdx.y - cg * L2, c0 = cg * BT;
    const int sc = (s < N) ? s : N - 1;
    double ax[BT], ay[BT];
#pragma unroll
    for (int j = 0; j < BT; ++j) { ax[j] = 0.0; ay[j] = 0.0; }
    for (int a = 0; a < L1; ++a) {
        const int t = L2 * a + b;
        double2 x;
        if (!INV) {
            const double xr = vreal[((size_t)blockIdx.z * L + t) * N + sc];
            if (TWISTED) { const double2 th = theta[t]; x = make_double2(th.x * xr, th.y * xr); }
            else x = make_double2(xr, 0.0);
        } else {
            const double2 *h = nu + (size_t)blockIdx.z * K * N;
            if (t < K) x = h[(size_t)t * N + sc];
            else { const int km = TWISTED ? L - 1 - t : L - t; const double2 cc = h[(size_t)km * N + sc]; x = make_double2(cc.x, -cc.y); }
        }
#pragma unroll
        for (int j = 0; j < BT; ++j) {
            const int c = (c0 + j < L1) ? c0 + j : L1 - 1;
            const double2 ww = W1[(size_t)c * L1 + a];
            const double wy = INV ? -ww.y : ww.y;
            ax[j] += ww.x * x.x - wy * x.y;
            ay[j] += ww.x * x.y + wy * x.x;
        }
    }
    const size_t base = (size_t)blockIdx.z * L * N;
#pragma unroll
    for (int j = 0; j < BT; ++j) {
        const int c = c0 + j;
        if (c < L1 && s < N) {
            const double2 tw = TW[(int)(((long long)c * b) % L)];
            const double ty = INV ? -tw.y : tw.y;
            Z[base + (size_t)(c * L2 + b) * N + s] = make_double2(tw.x * ax[j] - ty * ay[j], tw.x * ay[j] + ty * ax[j]);
        }
    }
}

// step 2:  X[c + L1 d] = scale sum_b W2[d][b]^(+-1) Z[c][b]   for d = d0 .. d0 + BT - 1
//   forward: written to the half spectrum nu[k], k = c + L1 d < K (rows beyond it are skipped);  inverse: out[t] = Re((TWISTED ? conj(Theta_t) : 1) X[t]),
//   t = c + L1 d, and — rz_part — the partial sums of rvec . out over this wave's sites in slot t * gridDim.x + site tile
// grid: x = site tile, y = (d group) * L1 + c, z = vector
template <bool INV, bool TWISTED>
__global__ void __launch_bounds__(WAVE) k_big_s2(double2 *__restrict__ nu, double *__restrict__ out, const double2 *__restrict__ Z,
                                                 const double2 *__restrict__ W2, const double2 *__restrict__ theta, int N, int L, int L1, int L2, int K,
                                                 double scale, const double *__restrict__ rvec, double *__restrict__ rz_part, int nrz) {
    const int s = blockIdx.x * WAVE + threadIdx.x;
    const int dg = blockIdx.y / L1, c = blockIdx.y - dg * L1, d0 = dg * BT;
    if (!INV && c + L1 * d0 >= K) return;                  // forward: the whole group lies beyond the half spectrum
    const int sc = (s < N) ? s : N - 1;
    const size_t base = (size_t)blockIdx.z * L * N;
    double ax[BT], ay[BT];
#pragma unroll
    for (int j = 0; j < BT; ++j) { ax[j] = 0.0; ay[j] = 0.0; }
    for (int b = 0; b < L2; ++b) {
        const double2 x = Z[base + (size_t)(c * L2 + b) * N + sc];
#pragma unroll
        for (int j = 0; j < BT; ++j) {
            const int d = (d0 + j < L2) ? d0 + j : L2 - 1;
            const double2 ww = W2[(size_t)d * L2 + b];
            const double wy = INV ? -ww.y : ww.y;
            ax[j] += ww.x * x.x - wy * x.y;
            ay[j] += ww.x * x.y + wy * x.x;
        }
    }
#pragma unroll
    for (int j = 0; j < BT; ++j) {
        const int d = d0 + j, k = c + L1 * d;
        if (d >= L2) continue;
        if (!INV) {
            if (k < K && s < N) nu[((size_t)blockIdx.z * K + k) * N + s] = make_double2(scale * ax[j], scale * ay[j]);
        } else {
            const double2 th = TWISTED ? theta[k] : make_double2(1.0, 0.0);
            const double vx = scale * ax[j], vy = scale * ay[j];
            const double o = th.x * vx + th.y * vy;        // Re(conj(theta) v)
            double dot = 0.0;
            if (s < N) {
                out[base + (size_t)k * N + s] = o;
                if (rz_part) dot = rvec[base + (size_t)k * N + s] * o;
            }
            if (rz_part) {
#pragma unroll
                for (int o2 = 32; o2 > 0; o2 >>= 1) dot += __shfl_xor(dot, o2, WAVE);
                if (threadIdx.x == 0) {
                    const int nst = (int)gridDim.x, slot = k * nst + (int)blockIdx.x, used = L * nst;
                    double *slots = rz_part + (size_t)blockIdx.z * nrz;
                    slots[slot] = dot;
                    for (int q = used + slot; q < nrz; q += used) slots[q] = 0.0;
                }
            }
        }
    }
}

int ensure_work(elph_handle_s *h, int N, int nvec) {
    const size_t need = (size_t)nvec * (size_t)h->L * (size_t)N;
    if (need <= h->big_cap) return ELPH_OK;
    HIPCHK(hipStreamSynchronize(h->stream));
    if (h->d_big_a) { HIPCHK(hipFree(h->d_big_a)); h->d_big_a = nullptr; }
    if (h->d_big_b) { HIPCHK(hipFree(h->d_big_b)); h->d_big_b = nullptr; }
    HIPCHK(hipMalloc((void **)&h->d_big_a, need * sizeof(double2)));
    HIPCHK(hipMalloc((void **)&h->d_big_b, need * sizeof(double2)));
    h->big_cap = need;
    return ELPH_OK;
}

// full complex transform of nvec vectors of N columns: result in h->d_big_a (input in h->d_big_a, scratch h->d_big_b)
template <bool INV>
int big_fft(elph_handle_s *h, int N, int nvec) {
    const int L = (int)h->L, L1 = h->big_L1, L2 = h->big_L2, nst = (N + WAVE - 1) / WAVE;
    const dim3 grid((unsigned)nst, (unsigned)L, (unsigned)nvec);
    if (L2 == 1) {                                        // no split: direct transform, result back into d_big_a
        hipLaunchKernelGGL((k_big_direct<INV>), grid, dim3(WAVE), 0, h->stream, h->d_big_b, h->d_big_a, h->d_big_TW, N, L, INV ? 1.0 / (double)L : 1.0);
        hipError_t e = hipMemcpyAsync(h->d_big_a, h->d_big_b, (size_t)nvec * L * N * sizeof(double2), hipMemcpyDeviceToDevice, h->stream);
        if (e != hipSuccess) { elph_set_error("dft_big: copy: %s", hipGetErrorString(e)); return ELPH_E_HIP; }
        return big_check(INV ? "k_big_direct(inverse)" : "k_big_direct(forward)");
    }
    hipLaunchKernelGGL((k_big_step1<INV>), grid, dim3(WAVE), 0, h->stream, h->d_big_b, h->d_big_a, h->d_big_W1, h->d_big_TW, N, L, L1, L2);
    hipLaunchKernelGGL((k_big_step2<INV>), grid, dim3(WAVE), 0, h->stream, h->d_big_a, h->d_big_b, h->d_big_W2, N, L, L1, L2,
                       INV ? 1.0 / (double)L : 1.0);
    return big_check(INV ? "big_fft(inverse)" : "big_fft(forward)");
}

}  // namespace

bool elph_dft_big(const elph_handle_s *h) { return h->big_L1 > 0; }

// host: factor L and build W1, W2, TW, Theta (exact index reduction, as in dft.hip)
int elph_dft_big_build_tables(elph_handle_s *h) {
    const int L = (int)h->L;
    h->big_L1 = h->big_L2 = 0;
    int best = 0;
    for (int f = 2; (long long)f * f <= L; ++f)
        if (L % f == 0 && L / f <= 1024) best = f;            // the largest divisor <= sqrt(L) whose cofactor fits
    // no such split (a prime, or a prime factor beyond 1024): L1 = L, L2 = 1 selects the direct transform (k_big_direct)
    const bool split = best >= 2;
    const int L1 = split ? best : L, L2 = split ? L / best : 1;
    std::vector<double2> W1(split ? (size_t)L1 * L1 : 1), W2((size_t)L2 * L2), TW((size_t)L), TH((size_t)L);
    for (int c = 0; split && c < L1; ++c)
        for (int a = 0; a < L1; ++a) { const double x = 2.0 * M_PI * (double)(((long long)c * a) % L1) / (double)L1; W1[(size_t)c * L1 + a] = make_double2(cos(x), -sin(x)); }
    for (int d = 0; d < L2; ++d)
        for (int b = 0; b < L2; ++b) { const double x = 2.0 * M_PI * (double)(((long long)d * b) % L2) / (double)L2; W2[(size_t)d * L2 + b] = make_double2(cos(x), -sin(x)); }
    for (int n = 0; n < L; ++n) {
        const double x = 2.0 * M_PI * (double)n / (double)L, y = M_PI * (double)n / (double)L;
        TW[(size_t)n] = make_double2(cos(x), -sin(x));
        TH[(size_t)n] = make_double2(cos(y), -sin(y));        // Theta_t = exp(-i pi t / L)
    }
    struct { double2 **d; std::vector<double2> *v; } tabs[] = {{&h->d_big_W1, &W1}, {&h->d_big_W2, &W2}, {&h->d_big_TW, &TW}, {&h->d_big_TH, &TH}};
    for (auto &tb : tabs) {
        if (*tb.d) { HIPCHK(hipFree(*tb.d)); *tb.d = nullptr; }
        HIPCHK(hipMalloc((void **)tb.d, tb.v->size() * sizeof(double2)));
        HIPCHK(hipMemcpy(*tb.d, tb.v->data(), tb.v->size() * sizeof(double2), hipMemcpyHostToDevice));
    }
    h->big_L1 = L1; h->big_L2 = L2;
    return ELPH_OK;
}

void elph_dft_big_free(elph_handle_s *h) {
    double2 **p[] = {&h->d_big_W1, &h->d_big_W2, &h->d_big_TW, &h->d_big_TH, &h->d_big_a, &h->d_big_b};
    for (auto q : p) if (*q) { (void)hipFree(*q); *q = nullptr; }
    h->big_cap = 0; h->big_L1 = h->big_L2 = 0;
}

// nu[vec][k][s], k < K: K = ceil(L/2) (twisted) or L/2 + 1 (plain)
// Which form: the blocked pair has BT times fewer waves — a batch of one or two right-hand sides does not fill the chip with them and is latency-bound
// (measured, profiles/r06/long_time_axes_blocked_transforms.log: 16 x 16 sites, 1000 slices, ONE right-hand side 29 -> 40 us per transform; 16: 224 -> 113;
// 64: 1152 -> 358), so it runs from 16384 one-row waves on.  ELPH_DFT_BIG_BLOCKED=0 / 1: never / always (A/B, tests; read per call)
static bool blocked_form(const elph_handle_s *h, int N, int nvec) {
    if (h->big_L2 <= 1) return false;
    const char *e = getenv("ELPH_DFT_BIG_BLOCKED");
    if (e) return e[0] != '0';
    return (long long)nvec * ((N + WAVE - 1) / WAVE) * h->L >= 16384;
}

int elph_dft_big_fwd(elph_handle_s *h, bool twisted, double2 *nu, const double *vS, int N, int nvec) {
    const int L = (int)h->L, K = twisted ? (L + 1) / 2 : L / 2 + 1;
    RC(ensure_work(h, N, nvec));
    if (blocked_form(h, N, nvec)) {
        const int L1 = h->big_L1, L2 = h->big_L2, nst = (N + WAVE - 1) / WAVE;
        const dim3 g1((unsigned)nst, (unsigned)(((L1 + BT - 1) / BT) * L2), (unsigned)nvec), g2((unsigned)nst, (unsigned)(((L2 + BT - 1) / BT) * L1), (unsigned)nvec);
        if (twisted) {
            hipLaunchKernelGGL((k_big_s1<false, true>), g1, dim3(WAVE), 0, h->stream, h->d_big_b, vS, (const double2 *)nullptr, h->d_big_W1, h->d_big_TW, h->d_big_TH, N, L, L1, L2, K);
            hipLaunchKernelGGL((k_big_s2<false, true>), g2, dim3(WAVE), 0, h->stream, nu, (double *)nullptr, h->d_big_b, h->d_big_W2, h->d_big_TH, N, L, L1, L2, K, 1.0, (const double *)nullptr, (double *)nullptr, 0);
        } else {
            hipLaunchKernelGGL((k_big_s1<false, false>), g1, dim3(WAVE), 0, h->stream, h->d_big_b, vS, (const double2 *)nullptr, h->d_big_W1, h->d_big_TW, h->d_big_TH, N, L, L1, L2, K);
            hipLaunchKernelGGL((k_big_s2<false, false>), g2, dim3(WAVE), 0, h->stream, nu, (double *)nullptr, h->d_big_b, h->d_big_W2, h->d_big_TH, N, L, L1, L2, K, 1.0, (const double *)nullptr, (double *)nullptr, 0);
        }
        return big_check("k_big_s1/s2(forward)");
    }
    const long long total = (long long)nvec * L * N;
    if (twisted) hipLaunchKernelGGL((k_big_load<true>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, h->stream, h->d_big_a, vS, h->d_big_TH, N, L, total);
    else hipLaunchKernelGGL((k_big_load<false>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, h->stream, h->d_big_a, vS, h->d_big_TH, N, L, total);
    RC(big_fft<false>(h, N, nvec));
    const long long th = (long long)nvec * K * N;
    hipLaunchKernelGGL(k_big_take, dim3((unsigned)((th + 255) / 256)), dim3(256), 0, h->stream, nu, h->d_big_a, N, L, K, th);
    return big_check("k_big_take");
}

int elph_dft_big_inv(elph_handle_s *h, bool twisted, double *outS, const double2 *nu, int N, int nvec, const double *rvec,
                     double *rz_part, int nrz) {
    const int L = (int)h->L, K = twisted ? (L + 1) / 2 : L / 2 + 1;
    if (rz_part && nrz < L) { elph_set_error("dft_big: %d partial slots needed, %d available", L, nrz); return ELPH_E_STATE; }
    RC(ensure_work(h, N, nvec));
    {
        const int nst = (N + WAVE - 1) / WAVE;
        if (blocked_form(h, N, nvec) && (!rz_part || nrz >= L * nst)) {
            const int L1 = h->big_L1, L2 = h->big_L2;
            const dim3 g1((unsigned)nst, (unsigned)(((L1 + BT - 1) / BT) * L2), (unsigned)nvec), g2((unsigned)nst, (unsigned)(((L2 + BT - 1) / BT) * L1), (unsigned)nvec);
            const double scale = 1.0 / (double)L;
            if (twisted) {
                hipLaunchKernelGGL((k_big_s1<true, true>), g1, dim3(WAVE), 0, h->stream, h->d_big_b, (const double *)nullptr, nu, h->d_big_W1, h->d_big_TW, h->d_big_TH, N, L, L1, L2, K);
                hipLaunchKernelGGL((k_big_s2<true, true>), g2, dim3(WAVE), 0, h->stream, (double2 *)nullptr, outS, h->d_big_b, h->d_big_W2, h->d_big_TH, N, L, L1, L2, K, scale, rvec, rz_part, nrz);
            } else {
                hipLaunchKernelGGL((k_big_s1<true, false>), g1, dim3(WAVE), 0, h->stream, h->d_big_b, (const double *)nullptr, nu, h->d_big_W1, h->d_big_TW, h->d_big_TH, N, L, L1, L2, K);
                hipLaunchKernelGGL((k_big_s2<true, false>), g2, dim3(WAVE), 0, h->stream, (double2 *)nullptr, outS, h->d_big_b, h->d_big_W2, h->d_big_TH, N, L, L1, L2, K, scale, rvec, rz_part, nrz);
            }
            return big_check("k_big_s1/s2(inverse)");
        }
    }
    const long long total = (long long)nvec * L * N;
    if (twisted) hipLaunchKernelGGL((k_big_expand<true>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, h->stream, h->d_big_a, nu, N, L, K, total);
    else hipLaunchKernelGGL((k_big_expand<false>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, h->stream, h->d_big_a, nu, N, L, K, total);
    RC(big_fft<true>(h, N, nvec));
    const dim3 grid((unsigned)L, (unsigned)nvec);
    if (twisted) hipLaunchKernelGGL((k_big_store<true>), grid, dim3(WAVE), 0, h->stream, outS, h->d_big_a, h->d_big_TH, N, L, rvec, rz_part, nrz);
    else hipLaunchKernelGGL((k_big_store<false>), grid, dim3(WAVE), 0, h->stream, outS, h->d_big_a, h->d_big_TH, N, L, rvec, rz_part, nrz);
    return big_check("k_big_store");
}
