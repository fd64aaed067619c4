// dft_mfma.hip — batched tau-axis transforms as dense f64 GEMMs on the matrix cores (v_mfma_f64_16x16x4_f64).
//
// Same transforms as dft.hip (TimeFreqFFTs.jl:55-73,112-130 twisted; FourierAcceleration.jl:91-143 plain): for a batch
// of right-hand sides the direct DFT  out[m][col] = sum_j W[m][j] * in[j][col]  is a (2K x L) x (L x N*nrhs) real GEMM
// (forward: rows = (k, re/im), j = tau; inverse: rows = tau, j = (k, re/im) with Hermitian weights and 1/L folded in).
// The scalar-twiddle kernels of dft.hip are SGPR-bound in a batch (each FMA pair needs a fresh 16-byte scalar load,
// 106 SGPRs hold < 30 twiddles: ~8 TFLOP/s); here a wave keeps its 16 data columns in registers for the whole
// reduction axis (one f64 per lane per 4x16 B tile), streams pre-swizzled 16x4 A tiles of W with one coalesced 512-byte
// load each, and issues MG independent MFMAs per A-tile step.  The single-solve (latency-bound) case stays on dft.hip.
//
// MFMA operand maps (MI355X_MICROARCH.md, f64 16x16x4): A[row = lane&15][j = lane>>4], B[j = lane>>4][col = lane&15],
// C/D reg r: [row = (lane>>4) + 4r][col = lane&15].  Forward tiles order their 16 rows so that a lane's four results are
// (re, im) of two adjacent frequencies: row rho = r0 + 4r  <->  k = 8*mt + 2*r0 + (r>>1), part = r&1  -> 16-byte stores.

#include <cmath>
#include <cstdlib>
#include <vector>

#include "cg_fast_common.h"      // reduce_partials_lane: the SAME summation order as k_cg_ap takes for r.z (PxFuse)

namespace {

typedef double double4_t __attribute__((ext_vector_type(4)));

constexpr int MG = 5;                 // row tiles (independent accumulators) per wave
constexpr int CW = 4;                 // waves per workgroup = adjacent column tiles walking the SAME A-tile stream: their
                                      // requests for a tile meet in the CU's L1 instead of each going to L2 (every wave
                                      // of a row group reads the same table lines — single-wave workgroups hot-spot a
                                      // few L2 channels)
constexpr int NT_CAND[6] = {12, 20, 32, 40, 44, 64};

__device__ __forceinline__ bool mf_done(const CgState *state, int rhs) {
    if (!state) return false;
    return __hip_atomic_load(&state[2 * rhs].done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
}

// grid: x = column tile (16 columns), y = row-tile group (MG tiles), z = right-hand side
template <int NT, bool INV>
__global__ void __launch_bounds__(CW * WAVE) k_dft_mfma(double *__restrict__ out, const double *__restrict__ in,
                                                   const double *__restrict__ W, int N, int L, int K, const CgState *state,
                                                   const double *__restrict__ rvec, double *__restrict__ rz_part, int nrz) {
    const int rhs = blockIdx.z;
    if (mf_done(state, rhs)) return;
    const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x >> 6, col = lane & 15, jj = lane >> 4;
    const int ctile = blockIdx.x * CW + wv;
    if (ctile * 16 >= N) return;                                         // no block-level sync below
    const int s = ctile * 16 + col;
    const int sc = (s < N) ? s : N - 1;
    // ---- the wave's 16 data columns over the whole reduction axis: b[tt] = in[j = 4 tt + jj][col]
    double b[NT];
    if (!INV) {
        const double *v = in + (size_t)rhs * N * L;                       // real [tau][site]
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
            int t = 4 * tt + jj;
            t = (t < L) ? t : L - 1;                                      // W is zero there; the clamp keeps the value finite
            b[tt] = v[(size_t)t * N + sc];
        }
    } else {
        const double *nu = in + (size_t)rhs * K * N * 2;                  // complex [k][site] as (re, im) doubles
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
            int k = 2 * tt + (jj >> 1);
            k = (k < K) ? k : K - 1;
            b[tt] = nu[((size_t)k * N + sc) * 2 + (jj & 1)];
        }
    }
    const int mt0 = blockIdx.y * MG;
    const double *Wg = W + ((size_t)mt0 * NT) * WAVE + lane;              // tile (mt, tt) at ((mt*NT + tt)*64 + lane)
    double4_t acc[MG];
#pragma unroll
    for (int g = 0; g < MG; ++g) acc[g] = (double4_t){0.0, 0.0, 0.0, 0.0};
    // A tiles come from L2 (the table is shared by every wave): a step's MFMAs last ~130 ns, an L2 round trip several times
    // that, so the tile stream runs PF steps ahead in a register ring (indices are compile-time after unrolling).
    constexpr int PF = 6;
    double a[PF + 1][MG];
#pragma unroll
    for (int p = 0; p < PF; ++p) {
        if (p < NT) {
#pragma unroll
            for (int g = 0; g < MG; ++g) a[p][g] = Wg[(size_t)(g * NT + p) * WAVE];
        }
    }
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) {
        if (tt + PF < NT) {
#pragma unroll
            for (int g = 0; g < MG; ++g) a[(tt + PF) % (PF + 1)][g] = Wg[(size_t)(g * NT + tt + PF) * WAVE];
        }
#pragma unroll
        for (int g = 0; g < MG; ++g) acc[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[tt % (PF + 1)][g], b[tt], acc[g], 0, 0, 0);
    }
    // ---- results
    const int r0 = lane >> 4;
    if (!INV) {
        double2 *o = reinterpret_cast<double2 *>(out) + (size_t)rhs * K * N;
#pragma unroll
        for (int g = 0; g < MG; ++g) {
            const int k = 8 * (mt0 + g) + 2 * r0;
            if (s < N) {
                if (k < K) o[(size_t)k * N + s] = make_double2(acc[g].x, acc[g].y);
                if (k + 1 < K) o[(size_t)(k + 1) * N + s] = make_double2(acc[g].z, acc[g].w);
            }
        }
    } else {
        double dot = 0.0;
#pragma unroll
        for (int g = 0; g < MG; ++g) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int t = 16 * (mt0 + g) + r0 + 4 * r;
                if (s < N && t < L) {
                    const size_t i = (size_t)rhs * N * L + (size_t)t * N + s;
                    const double val = acc[g][r];
                    out[i] = val;
                    if (rz_part) dot += rvec[i] * val;
                }
            }
        }
        if (rz_part) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) dot += __shfl_xor(dot, o, WAVE);
            if (lane == 0) {
                const int nct = (N + 15) / 16;
                const int G = nct * (int)gridDim.y, bid = (int)blockIdx.y * nct + ctile;
                double *slots = rz_part + (size_t)rhs * nrz;
                slots[bid] = dot;
                for (int q = G + bid; q < nrz; q += G) slots[q] = 0.0;
            }
        }
    }
}


// The same GEMM for ONE or a few right-hand sides (the reference's real call shape): one 16 x 16 output tile per wave, so a
// transform is 160 waves at config C instead of 32, each with ALL its operands requested up front (NT data values and NT
// pre-swizzled A tiles per lane: one memory round trip) and a dependent chain of NT MFMAs (40 x 32 cycles = 0.5 us).  The
// scalar-twiddle kernels of dft.hip take 9.1 / 6.3 us here (forward / inverse, SGPR-bound), k_dft_mfma above with its five row
// tiles per wave and its 6-deep tile ring more.  grid: x = column tiles / CW, y = row tile, z = right-hand side.
constexpr int CW1 = 1;                // waves per workgroup of k_dft_mfma_1 (one: 160 workgroups spread over the chip)
template <int NT, bool INV>
__global__ void __launch_bounds__(CW1 * WAVE) k_dft_mfma_1(double *__restrict__ out, const double *__restrict__ in,
                                                     const double *__restrict__ W, int N, int L, int K, const CgState *state,
                                                     const double *__restrict__ rvec, double *__restrict__ rz_part, int nrz) {
    const int rhs = blockIdx.z;
    // (the "finished" flag is asked for first and looked at after the operand loads have gone out: its round trip overlaps theirs)
    const int done_flag = state ? __hip_atomic_load(&state[2 * rhs].done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x >> 6, col = lane & 15, jj = lane >> 4;
    const int ctile = blockIdx.x * CW1 + wv;
    if (ctile * 16 >= N) return;
    const int s = ctile * 16 + col;
    const int sc = (s < N) ? s : N - 1;
    const int mt = blockIdx.y;
    double a[NT], b[NT];
    const double *Wg = W + ((size_t)mt * NT) * WAVE + lane;
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) a[tt] = Wg[(size_t)tt * WAVE];
    if (!INV) {
        const double *v = in + (size_t)rhs * N * L;
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
            int t = 4 * tt + jj;
            t = (t < L) ? t : L - 1;
            b[tt] = v[(size_t)t * N + sc];
        }
    } else {
        const double *nu = in + (size_t)rhs * K * N * 2;
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
            int k = 2 * tt + (jj >> 1);
            k = (k < K) ? k : K - 1;
            b[tt] = nu[((size_t)k * N + sc) * 2 + (jj & 1)];
        }
    }
    if (done_flag) return;
    // two accumulators (even / odd reduction tiles): the 40 MFMAs are a dependent chain of 32 cycles each otherwise
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
        double2 *o = reinterpret_cast<double2 *>(out) + (size_t)rhs * K * N;
        const int k = 8 * mt + 2 * r0;
        if (s < N) {
            if (k < K) o[(size_t)k * N + s] = make_double2(acc.x, acc.y);
            if (k + 1 < K) o[(size_t)(k + 1) * N + s] = make_double2(acc.z, acc.w);
        }
    } else {
        double dot = 0.0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int t = 16 * mt + r0 + 4 * r;
            if (s < N && t < L) {
                const size_t i = (size_t)rhs * N * L + (size_t)t * N + s;
                out[i] = acc[r];
                if (rz_part) dot += rvec[i] * acc[r];
            }
        }
        if (rz_part) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) dot += __shfl_xor(dot, o, WAVE);
            if (lane == 0) {
                const int nct = (N + 15) / 16;
                const int G = nct * (int)gridDim.y, bid = (int)blockIdx.y * nct + ctile;
                double *slots = rz_part + (size_t)rhs * nrz;
                slots[bid] = dot;
                for (int q = G + bid; q < nrz; q += G) slots[q] = 0.0;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Twisted transform, one even/odd-tau split (L even; H = L/2, Q = ceil(H/2) independent frequencies):
//   forward   nu_k = A_k + w_k B_k,  A (B) = half-length twisted DFT of the even (odd) time slices, w_k = e^{-i pi (2k+1)/L};
//             A_{H-1-k} = conj(A_k) for real data, so only k < Q is transformed: two (H x H) real GEMMs that share every A tile
//             instead of one (L x L) — half the MFMAs, half the table traffic, and a wave produces all frequencies of its 16
//             columns (the input is read once, not once per row group).
//   inverse   v_{2j+p} = (2/L) Re sum_{k<Q} d^p_k e^{i pi (2k+1) j / H},  d^p_k = c_k + conj(c_{H-1-k}),  c_k = nu_k conj(w_k)^p
// Row / reduction orderings are those of k_dft_mfma with K -> Q, L -> H.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int R2_NT_CAND[8] = {5, 10, 15, 20, 25, 32, 40, 50};       // reduction tiles: L/8 rounded up (L <= 400)

// XR (forward only): the CG residual update is done on the way in — r <- r - alpha (A p), alpha = rho / sum(p.Ap partials), the
// r.r partials of the stop test and alpha[rhs] for the next k_cg_ap — i.e. k_cg_xr folded into the transform that reads r anyway
// (one pass over r less and one launch less per preconditioned iteration).  Same arithmetic as k_cg_xr_fast; the r.r partials
// are per column tile instead of per time slice.
struct XrFuse {
    const double *z;          // A p
    double *r;                // residual, updated in place (the transform's input)
    const double *pap;        // p.Ap partials [nrhs][npap]
    double *rr;               // r.r partial slots [nrhs][rr_slots]
    double *alpha;            // [nrhs]
    int npap, rr_slots;
    // FOLD (forward, streaming form): a frequency whose Chebyshev series has order 1 is its leading coefficient — z_w = |c0|^2 r_w — so
    // the transform writes the spectrum of those frequencies already scaled and adds their share of r.(P^-1 r) (Parseval), and the
    // Chebyshev kernel does not touch them (config C: 47 of 80 frequencies, 111 MB of its 189 MB per iteration of 288 right-hand sides).
    // fold[chain][w][2] = {scale, weight of |nu_w|^2 in r.z}; {1, 0} for the frequencies the Chebyshev kernel keeps.
    const double *fold;
    double *frz;              // r.z partial slots [nrhs][fnrz]; this kernel fills slots fslot0 + column tile
    int fnch, fnrz, fslot0;
};

// PX (inverse only, streaming form, one row group): the tail of the preconditioned CG iteration is done on the way out — with
// z = P^-1 r in the accumulators,  x <- x + alpha p  (the pending update of this iteration) and  p <- z + beta p,  beta = (r.z) / rho
// from the r.z partials the Chebyshev kernel left in frequency space and the rho of the current state copy.  z itself is never
// written; the next k_cg_ap_chunk<PX> reads p only.  Same arithmetic per element as k_cg_ap_chunk's own p- and x-update
// (sv + beta qv, xv + alpha qv) and the same summation order for r.z (reduce_partials_lane): the two forms give the same bits.
struct PxFuse {
    double *p;                // [nrhs][ndim], slot 0 of the ping-pong pair: updated in place
    double *x;                // [nrhs][ndim]
    const double *alpha;      // [nrhs] step length of this iteration (written by the forward transform's XrFuse)
    const double *rz;         // r.z partial slots [nrhs][nrz]
    int nrz;
};

template <int NT, bool INV, bool XR>
__global__ void __launch_bounds__(CW * WAVE) k_dft_mfma_r2(double *__restrict__ out, const double *__restrict__ in,
                                                      const double *__restrict__ W, const double2 *__restrict__ tw, int N, int L,
                                                      const CgState *state, const double *__restrict__ rvec,
                                                      double *__restrict__ rz_part, int nrz, XrFuse X) {
    const int rhs = blockIdx.z;
    if (mf_done(state, rhs)) return;
    const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x >> 6, col = lane & 15, jj = lane >> 4;
    const int ctile = blockIdx.x * CW + wv;
    if (ctile * 16 >= N) return;
    const int s = ctile * 16 + col;
    const int sc = (s < N) ? s : N - 1;
    const int H = L >> 1, Q = (H + 1) >> 1;      // Q independent frequencies; H odd: the middle one is its own mirror
    double b0[NT], b1[NT];
    if (!INV && XR) {
        double a = 0.0;
        for (int i = lane; i < X.npap; i += WAVE)
            a += __hip_atomic_load(X.pap + (size_t)rhs * X.npap + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, WAVE);
        const double rho = __hip_atomic_load(&state[2 * rhs].rho, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const double alpha = rho / a;
        double *rw = X.r + (size_t)rhs * N * L;
        const double *zz = X.z + (size_t)rhs * N * L;
        double acc = 0.0;
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
            const int j = 4 * tt + jj;
            b0[tt] = 0.0; b1[tt] = 0.0;
            if (j < H && s < N) {                                         // each element is touched by exactly one lane
                const size_t i0 = (size_t)(2 * j) * N + s, i1 = i0 + N;
                const double n0 = rw[i0] - alpha * zz[i0], n1 = rw[i1] - alpha * zz[i1];
                rw[i0] = n0; rw[i1] = n1;
                acc += n0 * n0;
                acc += n1 * n1;
                b0[tt] = n0; b1[tt] = n1;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, WAVE);
        if (lane == 0) {
            const int nct = (N + 15) / 16;
            double *slots = X.rr + (size_t)rhs * X.rr_slots;
            slots[ctile] = acc;
            for (int q = nct + ctile; q < X.rr_slots; q += nct) slots[q] = 0.0;
            if (ctile == 0) X.alpha[rhs] = alpha;
        }
    } else if (!INV) {
        const double *v = in + (size_t)rhs * N * L;
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
            int j = 4 * tt + jj;
            j = (j < H) ? j : H - 1;                                      // W is zero there
            b0[tt] = v[(size_t)(2 * j) * N + sc];
            b1[tt] = v[(size_t)(2 * j + 1) * N + sc];
        }
    } else {
        const double2 *nu = reinterpret_cast<const double2 *>(in) + (size_t)rhs * H * N;
        const int part = jj & 1;
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
            int k = 2 * tt + (jj >> 1);
            k = (k < Q) ? k : Q - 1;
            const int kc = H - 1 - k;
            const double2 a = nu[(size_t)k * N + sc], c = nu[(size_t)kc * N + sc], wk = tw[k], wc = tw[kc];
            // p = 0: d = a + conj(c);   p = 1: d = a conj(w_k) + conj(c conj(w_kc)),  conj(w) = (w.x, -w.y)
            const double a1x = a.x * wk.x + a.y * wk.y, a1y = a.y * wk.x - a.x * wk.y;
            const double c1x = c.x * wc.x + c.y * wc.y, c1y = c.y * wc.x - c.x * wc.y;
            b0[tt] = part ? (a.y - c.y) : (a.x + c.x);
            b1[tt] = part ? (a1y - c1y) : (a1x + c1x);
        }
    }
    const int mt0 = blockIdx.y * MG;
    const double *Wg = W + ((size_t)mt0 * NT) * WAVE + lane;
    double4_t acc0[MG], acc1[MG];
#pragma unroll
    for (int g = 0; g < MG; ++g) { acc0[g] = (double4_t){0.0, 0.0, 0.0, 0.0}; acc1[g] = (double4_t){0.0, 0.0, 0.0, 0.0}; }
    constexpr int PF = (NT < 4) ? NT : 4;
    double a[PF + 1][MG];
#pragma unroll
    for (int p = 0; p < PF; ++p) {
#pragma unroll
        for (int g = 0; g < MG; ++g) a[p][g] = Wg[(size_t)(g * NT + p) * WAVE];
    }
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) {
        if (tt + PF < NT) {
#pragma unroll
            for (int g = 0; g < MG; ++g) a[(tt + PF) % (PF + 1)][g] = Wg[(size_t)(g * NT + tt + PF) * WAVE];
        }
#pragma unroll
        for (int g = 0; g < MG; ++g) {
            acc0[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[tt % (PF + 1)][g], b0[tt], acc0[g], 0, 0, 0);
            acc1[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[tt % (PF + 1)][g], b1[tt], acc1[g], 0, 0, 0);
        }
    }
    const int r0 = lane >> 4;
    if (!INV) {
        double2 *o = reinterpret_cast<double2 *>(out) + (size_t)rhs * H * N;
#pragma unroll
        for (int g = 0; g < MG; ++g) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int k = 8 * (mt0 + g) + 2 * r0 + e;
                if (s < N && k < Q) {
                    const double ax = e ? acc0[g].z : acc0[g].x, ay = e ? acc0[g].w : acc0[g].y;
                    const double bx = e ? acc1[g].z : acc1[g].x, by = e ? acc1[g].w : acc1[g].y;
                    const int kc = H - 1 - k;
                    const double2 wk = tw[k], wc = tw[kc];
                    o[(size_t)k * N + s] = make_double2(ax + (wk.x * bx - wk.y * by), ay + (wk.x * by + wk.y * bx));
                    // conj(A) + w_kc conj(B)
                    o[(size_t)kc * N + s] = make_double2(ax + (wc.x * bx + wc.y * by), -ay + (wc.y * bx - wc.x * by));
                }
            }
        }
    } else {
        double dot = 0.0;
#pragma unroll
        for (int g = 0; g < MG; ++g) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = 16 * (mt0 + g) + r0 + 4 * r;
                if (s < N && j < H) {
                    const size_t i = (size_t)rhs * N * L + (size_t)(2 * j) * N + s;
                    const double v0 = acc0[g][r], v1 = acc1[g][r];
                    out[i] = v0;
                    out[i + N] = v1;
                    if (rz_part) dot += rvec[i] * v0 + rvec[i + N] * v1;
                }
            }
        }
        if (rz_part) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) dot += __shfl_xor(dot, o, WAVE);
            if (lane == 0) {
                const int nct = (N + 15) / 16;
                const int G = nct * (int)gridDim.y, bid = (int)blockIdx.y * nct + ctile;
                double *slots = rz_part + (size_t)rhs * nrz;
                slots[bid] = dot;
                for (int q = G + bid; q < nrz; q += G) slots[q] = 0.0;
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// Streaming form of k_dft_mfma_r2: the W panel of the row group (MG x NT tiles, 51 KB at L = 160) is staged ONCE per workgroup in
// LDS and the data tiles are prefetched PFB reduction steps ahead in a small register ring instead of being held for the whole
// axis.  In the form above a wave runs load phase -> MFMA phase -> store phase and only other waves overlap them (2 waves per
// SIMD: transform time = memory time + MFMA time); here the vector-memory queue carries nothing but the data stream, the
// A operands come from LDS (ds_read, lgkmcnt), so loads run under the MFMAs of the same wave, and the smaller register
// footprint admits a third wave per SIMD.
// ---------------------------------------------------------------------------------------------------------------------
// RZ (inverse only): fuse the time-domain r.z partial sums (needs this lane's slice of r); without it the inverse is a pure
// transform (the Chebyshev kernel delivered r.z in frequency space) and fits a third wave per SIMD
template <int NT, bool INV, bool XR, bool RZ, bool PX = false>
__global__ void __launch_bounds__(CW * WAVE) k_dft_mfma_r2s(double *__restrict__ out, const double *__restrict__ in,
                                                       const double *__restrict__ W, const double2 *__restrict__ tw, int N, int L,
                                                       const CgState *state, const double *__restrict__ rvec,
                                                       double *__restrict__ rz_part, int nrz, XrFuse X, PxFuse PXF = PxFuse{}) {
    static_assert(!PX || (INV && !RZ && !XR), "PxFuse rides on the pure inverse transform");
    extern __shared__ double Wl[];                                       // [MG][NT][64]
    const int rhs = blockIdx.z;
    if (mf_done(state, rhs)) return;                                     // uniform over the workgroup
    const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x >> 6, col = lane & 15, jj = lane >> 4;
    const int ctile = blockIdx.x * CW + wv;
    const bool active = ctile * 16 < N;
    const int s = ctile * 16 + col;
    const int sc = (s < N) ? s : N - 1;
    const int H = L >> 1, Q = (H + 1) >> 1;      // Q independent frequencies; H odd: the middle one is its own mirror
    const int mt0 = blockIdx.y * MG;
    constexpr int PFW = INV ? 4 : 8;     // prefetch depth in reduction steps; 4 keeps the inverse at 3 waves per SIMD (4 doubles + a twiddle per slot)
    constexpr int PFB = (NT < PFW) ? NT : PFW;
    constexpr int RAW = (INV || XR) ? 4 : 2;
    double raw[PFB + 1][RAW];
    double2 twr[PFB + 1];

    const double *v = in + (size_t)rhs * N * L;                                            // forward: real [tau][site]
    const double2 *nu = reinterpret_cast<const double2 *>(in) + (size_t)rhs * H * N;       // inverse: complex [k][site]
    double *rw = XR ? X.r + (size_t)rhs * N * L : nullptr;
    const double *zz = XR ? X.z + (size_t)rhs * N * L : nullptr;
    auto issue = [&](int tt, double (&dst)[RAW], double2 &twd) {
        if (!INV) {
            int j = 4 * tt + jj;
            j = (j < H) ? j : H - 1;
            const size_t i0 = (size_t)(2 * j) * N + sc;
            if (XR) { dst[0] = rw[i0]; dst[1] = rw[i0 + N]; dst[2] = zz[i0]; dst[3] = zz[i0 + N]; }
            else    { dst[0] = v[i0]; dst[1] = v[i0 + N]; }
        } else {
            int k = 2 * tt + (jj >> 1);
            k = (k < Q) ? k : Q - 1;
            const double2 a = nu[(size_t)k * N + sc], c = nu[(size_t)(H - 1 - k) * N + sc];
            dst[0] = a.x; dst[1] = a.y; dst[2] = c.x; dst[3] = c.y;
            twd = tw[k];
        }
    };
    if (active) {
#pragma unroll
        for (int p = 0; p < PFB; ++p) issue(p, raw[p], twr[p]);
    }
    // ---- the W panel of this row group -> LDS (all waves, also those without a column tile: they pass the barrier)
    {
        const double2 *src = reinterpret_cast<const double2 *>(W + (size_t)mt0 * NT * WAVE);
        double2 *dst = reinterpret_cast<double2 *>(Wl);
        for (int i = threadIdx.x; i < MG * NT * WAVE / 2; i += CW * WAVE) dst[i] = src[i];
        // (the fold table of this right-hand side's chain rides behind the panel: the epilogue reads it from LDS, not through L2)
        if (!INV && X.fold) {
            const double2 *fs = reinterpret_cast<const double2 *>(X.fold) + (size_t)(rhs % X.fnch) * H;
            double2 *fd = reinterpret_cast<double2 *>(Wl + (size_t)MG * NT * WAVE);
            for (int i = threadIdx.x; i < H; i += CW * WAVE) fd[i] = fs[i];
        }
    }
    double alpha = 0.0, acc = 0.0;
    if (!INV && XR) {
        double a = 0.0;
        for (int i = lane; i < X.npap; i += WAVE)
            a += __hip_atomic_load(X.pap + (size_t)rhs * X.npap + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, WAVE);
        alpha = __hip_atomic_load(&state[2 * rhs].rho, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) / a;
    }
    __syncthreads();
    if (!active) return;
    const double *Al = Wl + lane;
    // inverse with the fused r.z: this lane's r values (the rows its accumulators will hold) are fetched now, under the MFMAs,
    // instead of in the epilogue where nothing hides them
    double rv0[(INV && RZ) ? MG * 4 : 1], rv1[(INV && RZ) ? MG * 4 : 1];
    if (INV && RZ) {
#pragma unroll
        for (int g = 0; g < MG; ++g) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int j = 16 * (mt0 + g) + (lane >> 4) + 4 * r;
                j = (j < H) ? j : H - 1;
                const size_t i = (size_t)rhs * N * L + (size_t)(2 * j) * N + sc;
                rv0[g * 4 + r] = rvec[i];
                rv1[g * 4 + r] = rvec[i + N];
            }
        }
    }
    // PxFuse: beta and alpha of this right-hand side; p and x of the first row tile fetched now, under the MFMAs (the other
    // tiles one ahead in the epilogue)
    double px_beta = 0.0, px_alpha = 0.0;
    double pq[PX ? 2 : 1][PX ? 8 : 1], xq[PX ? 2 : 1][PX ? 8 : 1];
    double *pb = PX ? PXF.p + (size_t)rhs * N * L : nullptr, *xb = PX ? PXF.x + (size_t)rhs * N * L : nullptr;
    auto px_fetch = [&](int g, double (&pv)[PX ? 8 : 1], double (&xv)[PX ? 8 : 1]) {
#pragma unroll
        for (int r = 0; r < (PX ? 4 : 0); ++r) {
            int jr = 16 * (mt0 + g) + (lane >> 4) + 4 * r;
            jr = (jr < H) ? jr : H - 1;
            const size_t i = (size_t)(2 * jr) * N + sc;
            pv[2 * r] = pb[i]; pv[2 * r + 1] = pb[i + N];
            xv[2 * r] = xb[i]; xv[2 * r + 1] = xb[i + N];
        }
    };
    if constexpr (PX) {
        const double rzs = reduce_partials_lane(PXF.rz + (size_t)rhs * PXF.nrz, PXF.nrz, lane);
        px_beta = rzs / __hip_atomic_load(&state[2 * rhs].rho, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        px_alpha = __hip_atomic_load(PXF.alpha + rhs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        px_fetch(0, pq[0], xq[0]);
    }
    double4_t acc0[MG], acc1[MG];
#pragma unroll
    for (int g = 0; g < MG; ++g) { acc0[g] = (double4_t){0.0, 0.0, 0.0, 0.0}; acc1[g] = (double4_t){0.0, 0.0, 0.0, 0.0}; }
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) {
        if (tt + PFB < NT) issue(tt + PFB, raw[(tt + PFB) % (PFB + 1)], twr[(tt + PFB) % (PFB + 1)]);
        const double (&q)[RAW] = raw[tt % (PFB + 1)];
        double b0, b1;
        if (!INV && XR) {
            const int j = 4 * tt + jj;
            const bool valid = (j < H) && (s < N);                       // each element of r belongs to exactly one lane
            const double n0 = q[0] - alpha * q[2], n1 = q[1] - alpha * q[3];
            if (valid) {
                const size_t i0 = (size_t)(2 * j) * N + s;
                rw[i0] = n0; rw[i0 + N] = n1;
                acc += n0 * n0;
                acc += n1 * n1;
            }
            b0 = valid ? n0 : 0.0; b1 = valid ? n1 : 0.0;
        } else if (!INV) {
            b0 = q[0]; b1 = q[1];
        } else {
            // d0 = nu_k + conj(nu_kc);  d1 = conj(w_k) (nu_k - conj(nu_kc))   (w_kc = -conj(w_k))
            const double2 wk = twr[tt % (PFB + 1)];
            const double sx = q[0] + q[2], sy = q[1] - q[3], ex = q[0] - q[2], ey = q[1] + q[3];
            const double d1x = ex * wk.x + ey * wk.y, d1y = ey * wk.x - ex * wk.y;
            b0 = (jj & 1) ? sy : sx;
            b1 = (jj & 1) ? d1y : d1x;
        }
#pragma unroll
        for (int g = 0; g < MG; ++g) {
            const double a = Al[(size_t)(g * NT + tt) * WAVE];
            acc0[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b0, acc0[g], 0, 0, 0);
            acc1[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b1, acc1[g], 0, 0, 0);
        }
    }
    const int r0 = lane >> 4;
    if (!INV) {
        if (XR) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, WAVE);
            const int nct = (N + 15) / 16;
            if (nct <= X.rr_slots) {
                if (lane == 0) {
                    double *slots = X.rr + (size_t)rhs * X.rr_slots;
                    slots[ctile] = acc;
                    for (int qq = nct + ctile; qq < X.rr_slots; qq += nct) slots[qq] = 0.0;
                    if (ctile == 0) X.alpha[rhs] = alpha;
                }
            } else {
                // (round 6) more column tiles than r.r slots — a large lattice on a short time axis (32 x 32 at Ltau = 40: 64 tiles, 40 slots): ONE
                // slot per workgroup, the waves' sums added in wave order through LDS.  (Waves without a column tile have left: a barrier
                // waits for the surviving waves of the workgroup only; wave 0 always has a tile.)
                __shared__ double wsum[CW];
                if (lane == 0) wsum[wv] = acc;
                __syncthreads();
                if (wv == 0 && lane == 0) {
                    const int nw = (nct - (int)blockIdx.x * CW < CW) ? nct - (int)blockIdx.x * CW : CW, nwg = (int)gridDim.x;
                    double t = 0.0;
                    for (int w2 = 0; w2 < nw; ++w2) t += wsum[w2];
                    double *slots = X.rr + (size_t)rhs * X.rr_slots;
                    slots[blockIdx.x] = t;
                    for (int qq = nwg + (int)blockIdx.x; qq < X.rr_slots; qq += nwg) slots[qq] = 0.0;
                    if (blockIdx.x == 0) X.alpha[rhs] = alpha;
                }
            }
        }
        double facc = 0.0;
        const double *fch = Wl + (size_t)MG * NT * WAVE;          // (LDS copy of this chain's fold table, staged with the panel)
        double2 *o = reinterpret_cast<double2 *>(out) + (size_t)rhs * H * N;
#pragma unroll
        for (int g = 0; g < MG; ++g) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int k = 8 * (mt0 + g) + 2 * r0 + e;
                if (s < N && k < Q) {
                    const double ax = e ? acc0[g].z : acc0[g].x, ay = e ? acc0[g].w : acc0[g].y;
                    const double bx = e ? acc1[g].z : acc1[g].x, by = e ? acc1[g].w : acc1[g].y;
                    const double2 wk = tw[k];
                    const double tx = wk.x * bx - wk.y * by, ty = wk.x * by + wk.y * bx;          // t = w_k B
                    double2 n0 = make_double2(ax + tx, ay + ty);                                   // nu_k = A + t
                    double2 n1 = make_double2(ax - tx, -(ay - ty));                                // nu_kc = conj(A - t)
                    if (X.fold) {
                        const double2 f0 = reinterpret_cast<const double2 *>(fch)[k], f1 = reinterpret_cast<const double2 *>(fch)[H - 1 - k];
                        const double s0 = f0.x, w0 = f0.y, s1 = f1.x, w1 = f1.y;
                        facc += w0 * (n0.x * n0.x + n0.y * n0.y);
                        if (H - 1 - k != k) facc += w1 * (n1.x * n1.x + n1.y * n1.y);
                        n0.x *= s0; n0.y *= s0; n1.x *= s1; n1.y *= s1;
                    }
                    o[(size_t)k * N + s] = n0;
                    o[(size_t)(H - 1 - k) * N + s] = n1;
                }
            }
        }
        if (X.fold && blockIdx.y == 0) {      // (one row group: every frequency of these 16 columns is in this wave)
#pragma unroll
            for (int o2 = 32; o2 > 0; o2 >>= 1) facc += __shfl_xor(facc, o2, WAVE);
            if (lane == 0) X.frz[(size_t)rhs * X.fnrz + X.fslot0 + ctile] = facc;
        }
    } else if constexpr (PX) {
#pragma unroll
        for (int g = 0; g < MG; ++g) {
            if (g + 1 < MG) px_fetch(g + 1, pq[(g + 1) & 1], xq[(g + 1) & 1]);
            const double (&pv)[8] = pq[g & 1];
            const double (&xv)[8] = xq[g & 1];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = 16 * (mt0 + g) + r0 + 4 * r;
                if (s < N && j < H) {
                    const size_t i = (size_t)(2 * j) * N + s;
                    xb[i] = xv[2 * r] + px_alpha * pv[2 * r];
                    xb[i + N] = xv[2 * r + 1] + px_alpha * pv[2 * r + 1];
                    pb[i] = acc0[g][r] + px_beta * pv[2 * r];
                    pb[i + N] = acc1[g][r] + px_beta * pv[2 * r + 1];
                }
            }
        }
    } else {
        double dot = 0.0;
#pragma unroll
        for (int g = 0; g < MG; ++g) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = 16 * (mt0 + g) + r0 + 4 * r;
                if (s < N && j < H) {
                    const size_t i = (size_t)rhs * N * L + (size_t)(2 * j) * N + s;
                    const double v0 = acc0[g][r], v1 = acc1[g][r];
                    out[i] = v0;
                    out[i + N] = v1;
                    if (RZ) dot += rv0[g * 4 + r] * v0 + rv1[g * 4 + r] * v1;
                }
            }
        }
        if (RZ) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) dot += __shfl_xor(dot, o, WAVE);
            if (lane == 0) {
                const int nct = (N + 15) / 16;
                const int G = nct * (int)gridDim.y, bid = (int)blockIdx.y * nct + ctile;
                double *slots = rz_part + (size_t)rhs * nrz;
                slots[bid] = dot;
                for (int qq = G + bid; qq < nrz; qq += G) slots[qq] = 0.0;
            }
        }
    }
}

int pick_nt_r2(int need) {
    for (int c : R2_NT_CAND) if (need <= c) return c;
    return 0;
}

bool r2_enabled() {
    const char *e = getenv("ELPH_DFT_R2");         // read per call (tests compare both forms)
    return !(e && atoi(e) == 0);
}

int pick_nt(int need) {
    for (int c : NT_CAND) if (need <= c) return c;
    return 0;
}

int mf_check(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { elph_set_error("launch %s failed: %s", what, hipGetErrorString(e)); return ELPH_E_HIP; }
    return ELPH_OK;
}

template <bool INV>
int launch(elph_handle_s *h, int nt, double *out, const double *in, const double *W, int N, int K, int groups, int nrhs,
           const CgState *st, const double *rvec, double *rz_part, int nrz) {
    const int nct = (N + 15) / 16;
    const dim3 grid((unsigned)((nct + CW - 1) / CW), (unsigned)groups, (unsigned)nrhs), block(CW * WAVE);
    const int L = (int)h->L;
#define MF_CASE(NTV) case NTV: hipLaunchKernelGGL((k_dft_mfma<NTV, INV>), grid, block, 0, h->stream, out, in, W, N, L, K, st, rvec, rz_part, nrz); break;
    switch (nt) {
        MF_CASE(12) MF_CASE(20) MF_CASE(32) MF_CASE(40) MF_CASE(44) MF_CASE(64)
        default: elph_set_error("dft_mfma: no kernel for %d reduction tiles", nt); return ELPH_E_UNSUPPORTED;
    }
#undef MF_CASE
    return mf_check(INV ? "k_dft_mfma(inverse)" : "k_dft_mfma(forward)");
}

template <bool INV>
int launch_1(elph_handle_s *h, int nt, double *out, const double *in, const double *W, int N, int K, int row_tiles, int nrhs,
             const CgState *st, const double *rvec, double *rz_part, int nrz) {
    const int nct = (N + 15) / 16;
    const dim3 grid((unsigned)((nct + CW1 - 1) / CW1), (unsigned)row_tiles, (unsigned)nrhs), block(CW1 * WAVE);
    const int L = (int)h->L;
#define MF1_CASE(NTV) case NTV: hipLaunchKernelGGL((k_dft_mfma_1<NTV, INV>), grid, block, 0, h->stream, out, in, W, N, L, K, st, rvec, rz_part, nrz); break;
    switch (nt) {
        MF1_CASE(12) MF1_CASE(20) MF1_CASE(32) MF1_CASE(40) MF1_CASE(44) MF1_CASE(64)
        default: elph_set_error("dft_mfma_1: no kernel for %d reduction tiles", nt); return ELPH_E_UNSUPPORTED;
    }
#undef MF1_CASE
    return mf_check(INV ? "k_dft_mfma_1(inverse)" : "k_dft_mfma_1(forward)");
}

template <bool INV, bool XR = false>
int launch_r2(elph_handle_s *h, const elph_handle_s::MfmaTab &T, double *out, const double *in, int N, int nrhs, const CgState *st,
              const double *rvec, double *rz_part, int nrz, XrFuse X = XrFuse{}, PxFuse PXF = PxFuse{}) {
    const int nct = (N + 15) / 16;
    const dim3 grid((unsigned)((nct + CW - 1) / CW), (unsigned)T.groups, (unsigned)nrhs), block(CW * WAVE);
    const int L = (int)h->L;
    const double2 *tw = reinterpret_cast<const double2 *>(h->d_r2_tw);
    // streaming form: W panel of a row group in LDS.  Up to 64 KB is the default dynamic-LDS limit; the long time axes
    // (L = 256 ... 400: 80 ... 128 KB of the CU's 160 KB) ask for it explicitly, once per kernel instantiation
    const size_t panel = (size_t)MG * T.nt * WAVE * sizeof(double);
    const size_t shm_s = panel + (X.fold ? (size_t)(L / 2) * 2 * sizeof(double) : 0);      // + the fold table of the chain (forward, XrFuse)
    const char *es = getenv("ELPH_DFT_STREAM");
    if (panel <= 144 * 1024 && !(es && atoi(es) == 0)) {
        hipError_t attr_rc = hipSuccess;
#define R2S_LAUNCH(NTV, RZV, PXV) do {                                                                                         \
            auto kfn = k_dft_mfma_r2s<NTV, INV, XR, RZV, PXV>;                                                                  \
            if (shm_s > 64 * 1024) {                                                                                            \
                static bool raised = false;                                                                                     \
                if (!raised) { attr_rc = hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(panel + 8192)); raised = (attr_rc == hipSuccess); } \
            }                                                                                                                   \
            if (attr_rc == hipSuccess) hipLaunchKernelGGL(kfn, grid, block, shm_s, h->stream, out, in, T.W, tw, N, L, st, rvec, rz_part, nrz, X, PXF); \
        } while (0)
#define R2S_CASE(NTV) case NTV: if (INV && !XR && PXF.p) R2S_LAUNCH(NTV, false, (INV && !XR)); else if (INV && rz_part) R2S_LAUNCH(NTV, INV, false); else R2S_LAUNCH(NTV, false, false); break;
        switch (T.nt) {
            R2S_CASE(5) R2S_CASE(10) R2S_CASE(15) R2S_CASE(20) R2S_CASE(25) R2S_CASE(32) R2S_CASE(40) R2S_CASE(50)
            default: elph_set_error("dft_mfma_r2s: no kernel for %d reduction tiles", T.nt); return ELPH_E_UNSUPPORTED;
        }
#undef R2S_CASE
#undef R2S_LAUNCH
        if (attr_rc != hipSuccess) { elph_set_error("hipFuncSetAttribute(dynamic LDS %zu B) failed: %s", panel, hipGetErrorString(attr_rc)); return ELPH_E_HIP; }
        return mf_check(INV ? "k_dft_mfma_r2s(inverse)" : "k_dft_mfma_r2s(forward)");
    }
    if (PXF.p) { elph_set_error("dft_mfma: the p/x-fused inverse exists in the streaming form only"); return ELPH_E_STATE; }
#define R2_CASE(NTV) case NTV: hipLaunchKernelGGL((k_dft_mfma_r2<NTV, INV, XR>), grid, block, 0, h->stream, out, in, T.W, tw, N, L, st, rvec, rz_part, nrz, X); break;
    switch (T.nt) {
        R2_CASE(5) R2_CASE(10) R2_CASE(15) R2_CASE(20) R2_CASE(25) R2_CASE(32)
        default: elph_set_error("dft_mfma_r2: no register-resident kernel for %d reduction tiles (ELPH_DFT_STREAM=0 needs L <= 256)", T.nt); return ELPH_E_UNSUPPORTED;
    }
#undef R2_CASE
    return mf_check(INV ? "k_dft_mfma_r2(inverse)" : "k_dft_mfma_r2(forward)");
}

}  // namespace

// which: 0 twisted, 1 plain
bool elph_dft_mfma_usable(const elph_handle_s *h, int which, bool inverse, int N, int nrhs) {
    const char *e = getenv("ELPH_DFT_MFMA");       // read per call: tests switch both ways inside one process
    const int force = e ? atoi(e) : -1;
    if (force == 0) return false;
    const elph_handle_s::MfmaTab &T = h->mf[which][inverse ? 1 : 0];
    const bool split = which == 0 && h->mf_r2[inverse ? 1 : 0].W && r2_enabled();      // exists up to L = 400, the direct form to 256
    if (!T.W && !split) return false;
    if (force == 1) return true;
    // measured crossover against the scalar-twiddle kernels (tools/time_dft_crossover.py, configs C and D): the split form wins
    // from ~100 column-tile waves (8 right-hand sides at N = 256), the direct form from ~512 waves; against the
    // one-tile-per-wave form (k_dft_mfma_1; config C, preconditioned iteration: 54.8 vs 69.9 us at 10 right-hand sides, 71.8 vs
    // 74.2 at 24, 80.7 vs 76.9 at 32, 100.7 vs 83.5 at 48) from ~450
    if (split) return (long long)((N + 15) / 16) * nrhs >= (which == 0 && elph_dft_mfma1_usable(h, inverse, N, 0) ? 448 : 100);
    return (long long)((N + 15) / 16) * T.groups * nrhs >= 512;
}

// one output tile per wave: the twisted pair for batches below the crossover of the forms above (ELPH_DFT_MFMA1=0: scalar kernels)
bool elph_dft_mfma1_usable(const elph_handle_s *h, bool inverse, int N, int nrz_slots) {
    const char *e = getenv("ELPH_DFT_MFMA1");
    if (e && atoi(e) == 0) return false;
    const elph_handle_s::MfmaTab &T = h->mf[0][inverse ? 1 : 0];
    if (!T.W) return false;
    return !inverse || nrz_slots <= 0 || ((N + 15) / 16) * T.groups * MG <= nrz_slots;
}
int elph_dft_mfma1_fwd(elph_handle_s *h, double2 *nu, const double *vS, int N, int nrhs, const CgState *st) {
    const elph_handle_s::MfmaTab &T = h->mf[0][0];
    return launch_1<false>(h, T.nt, reinterpret_cast<double *>(nu), vS, T.W, N, (int)(h->L + 1) / 2, T.groups * MG, nrhs, st, nullptr, nullptr, 0);
}
int elph_dft_mfma1_inv(elph_handle_s *h, double *outS, const double2 *nu, int N, int nrhs, const CgState *st, const double *rvec,
                       double *rz_part, int nrz) {
    const elph_handle_s::MfmaTab &T = h->mf[0][1];
    return launch_1<true>(h, T.nt, outS, reinterpret_cast<const double *>(nu), T.W, N, (int)(h->L + 1) / 2, T.groups * MG, nrhs, st, rvec, rz_part, nrz);
}

int elph_dft_mfma_fwd(elph_handle_s *h, int which, double2 *nu, const double *vS, int N, int nrhs, const CgState *st) {
    if (which == 0 && h->mf_r2[0].W && r2_enabled())
        return launch_r2<false>(h, h->mf_r2[0], reinterpret_cast<double *>(nu), vS, N, nrhs, st, nullptr, nullptr, 0);
    const elph_handle_s::MfmaTab &T = h->mf[which][0];
    const int K = which == 0 ? (int)(h->L + 1) / 2 : (int)h->L / 2 + 1;
    return launch<false>(h, T.nt, reinterpret_cast<double *>(nu), vS, T.W, N, K, T.groups, nrhs, st, nullptr, nullptr, 0);
}

// whether launch_r2<false> takes the streaming form (the one that knows the order-1 fold)
bool elph_dft_mfma_fold_usable(const elph_handle_s *h) {
    const char *ef = getenv("ELPH_KPM_FOLD");
    if (ef && atoi(ef) == 0) return false;
    const elph_handle_s::MfmaTab &T = h->mf_r2[0];
    if (!T.W || !r2_enabled() || T.groups != 1 || (h->L & 1)) return false;
    const size_t panel = (size_t)MG * T.nt * WAVE * sizeof(double);
    const char *es = getenv("ELPH_DFT_STREAM");
    return panel <= 144 * 1024 && !(es && atoi(es) == 0);
}

// forward twisted transform of r - alpha z with the residual update of k_cg_xr folded in (see XrFuse); usable: see below
bool elph_dft_mfma_xr_usable(const elph_handle_s *h, int N, int nrhs) {
    const char *e = getenv("ELPH_FUSE_XR");
    if (e && atoi(e) == 0) return false;
    // r.r partial slots: one per column tile where the time axis has that many (L slots per right-hand side), else — streaming form only —
    // one per workgroup of CW tiles (round 6: large lattices on short time axes)
    const int nct = (N + 15) / 16;
    const size_t panel = (size_t)MG * h->mf_r2[0].nt * WAVE * sizeof(double);
    const char *es = getenv("ELPH_DFT_STREAM");
    const bool streaming = panel <= 144 * 1024 && !(es && atoi(es) == 0);
    const bool slots_ok = nct <= (int)h->L || (streaming && (nct + CW - 1) / CW <= (int)h->L);
    return h->mf_r2[0].W && r2_enabled() && elph_dft_mfma_usable(h, 0, false, N, nrhs) && slots_ok &&
           h->mf_r2[0].groups == 1;           // one row group: every element of r belongs to exactly one wave
}

int elph_dft_mfma_fwd_xr(elph_handle_s *h, double2 *nu, double *rS, const double *zS, const double *pap, int npap, double *rr,
                         double *alpha, int N, int nrhs, const CgState *st, const double *fold, int fold_nch, double *frz, int fnrz,
                         int fslot0) {
    XrFuse X{zS, rS, pap, rr, alpha, npap, (int)h->L, fold, frz, fold_nch, fnrz, fslot0};
    return launch_r2<false, true>(h, h->mf_r2[0], reinterpret_cast<double *>(nu), rS, N, nrhs, st, nullptr, nullptr, 0, X);
}

int elph_dft_mfma_inv(elph_handle_s *h, int which, double *outS, const double2 *nu, int N, int nrhs, const CgState *st,
                      const double *rvec, double *rz_part, int nrz) {
    if (which == 0 && h->mf_r2[1].W && r2_enabled()) {
        const elph_handle_s::MfmaTab &T2 = h->mf_r2[1];
        if (rz_part && (int)((N + 15) / 16) * T2.groups > nrz) { elph_set_error("dft_mfma_r2: %d partial slots needed, %d available", ((N + 15) / 16) * T2.groups, nrz); return ELPH_E_STATE; }
        return launch_r2<true>(h, T2, outS, reinterpret_cast<const double *>(nu), N, nrhs, st, rvec, rz_part, nrz);
    }
    const elph_handle_s::MfmaTab &T = h->mf[which][1];
    const int K = which == 0 ? (int)(h->L + 1) / 2 : (int)h->L / 2 + 1;
    if (rz_part && (int)((N + 15) / 16) * T.groups > nrz) { elph_set_error("dft_mfma: %d partial slots needed, %d available", ((N + 15) / 16) * T.groups, nrz); return ELPH_E_STATE; }
    return launch<true>(h, T.nt, outS, reinterpret_cast<const double *>(nu), T.W, N, K, T.groups, nrhs, st, rvec, rz_part, nrz);
}

// inverse twisted transform with the p- and x-update of the preconditioned CG iteration in its epilogue (PxFuse)
bool elph_dft_mfma_px_usable(const elph_handle_s *h, int N, int nrhs) {
    const char *e = getenv("ELPH_FUSE_PX");
    if (e && atoi(e) == 0) return false;
    const elph_handle_s::MfmaTab &T = h->mf_r2[1];
    if (!T.W || !r2_enabled() || T.groups != 1 || (h->L & 1)) return false;       // one row group: every element of p, x belongs to one lane
    const size_t panel = (size_t)MG * T.nt * WAVE * sizeof(double);
    const char *es = getenv("ELPH_DFT_STREAM");
    return panel <= 144 * 1024 && !(es && atoi(es) == 0) && elph_dft_mfma_usable(h, 0, true, N, nrhs);
}

int elph_dft_mfma_inv_px(elph_handle_s *h, const double2 *nu, int N, int nrhs, const CgState *st, double *pS, double *xS,
                         const double *alpha, const double *rz, int nrz) {
    if (!st) { elph_set_error("dft_mfma: the p/x-fused inverse needs the CG state"); return ELPH_E_STATE; }
    PxFuse PXF{pS, xS, alpha, rz, nrz};
    return launch_r2<true>(h, h->mf_r2[1], nullptr, reinterpret_cast<const double *>(nu), N, nrhs, st, nullptr, nullptr, 0, XrFuse{}, PXF);
}

// host: the four W matrices in A-tile order (zero-padded to groups*MG row tiles x nt reduction tiles)
int elph_dft_mfma_build_tables(elph_handle_s *h) {
    const int L = (int)h->L;
    for (int which = 0; which < 2; ++which) {
        const int K = which == 0 ? (L + 1) / 2 : L / 2 + 1;
        auto angle = [&](int k, int t) {   // exact index reduction, as in dft.hip
            if (which == 0) { const long long m = ((long long)(2 * k + 1) * t) % (2LL * L); return M_PI * (double)m / (double)L; }
            const long long m = ((long long)k * t) % L;
            return 2.0 * M_PI * (double)m / (double)L;
        };
        auto weight = [&](int k) {
            if (which == 0) return ((L & 1) && k == K - 1) ? 1.0 : 2.0;
            return (k == 0 || 2 * k == L) ? 1.0 : 2.0;
        };
        for (int inv = 0; inv < 2; ++inv) {
            elph_handle_s::MfmaTab &T = h->mf[which][inv];
            if (T.W) { HIPCHK(hipFree(T.W)); T.W = nullptr; }
            const int rows = inv ? L : 2 * K, red = inv ? 2 * K : L;
            const int nmt = (rows + 15) / 16, ntt = (red + 3) / 4;
            T.nt = pick_nt(ntt);
            if (T.nt == 0) continue;                       // L > 256: scalar kernels only
            T.groups = (nmt + MG - 1) / MG;
            const int nmt_pad = T.groups * MG;
            std::vector<double> W((size_t)nmt_pad * T.nt * WAVE, 0.0);
            for (int mt = 0; mt < nmt; ++mt)
                for (int tt = 0; tt < ntt; ++tt)
                    for (int lane = 0; lane < WAVE; ++lane) {
                        const int rho = lane & 15, jj = lane >> 4;
                        double val = 0.0;
                        if (!inv) {
                            const int r0 = rho & 3, r = rho >> 2;
                            const int k = 8 * mt + 2 * r0 + (r >> 1), part = r & 1, t = 4 * tt + jj;
                            if (k < K && t < L) { const double a = angle(k, t); val = part == 0 ? cos(a) : -sin(a); }
                        } else {
                            const int t = 16 * mt + rho, k = 2 * tt + (jj >> 1), part = jj & 1;
                            if (t < L && k < K) {
                                const double a = angle(k, t), w = weight(k) / (double)L;
                                val = part == 0 ? w * cos(a) : -(w * sin(a));     // Re((c + i s)(x + i y)) = c x - s y
                            }
                        }
                        W[((size_t)mt * T.nt + tt) * WAVE + lane] = val;
                    }
            HIPCHK(hipMalloc((void **)&T.W, W.size() * sizeof(double)));
            HIPCHK(hipMemcpy(T.W, W.data(), W.size() * sizeof(double), hipMemcpyHostToDevice));
        }
    }
    // ---- the even/odd split of the twisted transform (L even): half-length tables and the twiddles
    for (auto &T : h->mf_r2) if (T.W) { HIPCHK(hipFree(T.W)); T.W = nullptr; }
    if (h->d_r2_tw) { HIPCHK(hipFree(h->d_r2_tw)); h->d_r2_tw = nullptr; }
    if (L % 2 == 0 && L >= 8) {
        const int H = L / 2, Q = (H + 1) / 2;
        auto angle = [&](int k, int j) { const long long m = ((long long)(2 * k + 1) * j) % (2LL * H); return M_PI * (double)m / (double)H; };
        for (int inv = 0; inv < 2; ++inv) {
            elph_handle_s::MfmaTab &T = h->mf_r2[inv];
            // forward: rows (k < Q, re/im), 8 frequencies per row tile, reduction j < H; inverse: rows j < H, reduction (k < Q, re/im)
            const int nmt = inv ? (H + 15) / 16 : (Q + 7) / 8, ntt = inv ? (Q + 1) / 2 : (H + 3) / 4;
            T.nt = pick_nt_r2(ntt);
            if (T.nt == 0) continue;
            T.groups = (nmt + MG - 1) / MG;
            const int nmt_pad = T.groups * MG;
            std::vector<double> W((size_t)nmt_pad * T.nt * WAVE, 0.0);
            for (int mt = 0; mt < nmt; ++mt)
                for (int tt = 0; tt < ntt; ++tt)
                    for (int lane = 0; lane < WAVE; ++lane) {
                        const int rho = lane & 15, jj = lane >> 4;
                        double val = 0.0;
                        if (!inv) {
                            const int r0 = rho & 3, r = rho >> 2;
                            const int k = 8 * mt + 2 * r0 + (r >> 1), part = r & 1, j = 4 * tt + jj;
                            if (k < Q && j < H) { const double a = angle(k, j); val = part == 0 ? cos(a) : -sin(a); }
                        } else {
                            const int j = 16 * mt + rho, k = 2 * tt + (jj >> 1), part = jj & 1;
                            if (j < H && k < Q) {
                                // H odd: k = (H-1)/2 is its own mirror, its folded pair d = c + conj(c) counts the term twice
                                const double a = angle(k, j), w = ((H & 1) && k == Q - 1 ? 1.0 : 2.0) / (double)L;
                                val = part == 0 ? w * cos(a) : -(w * sin(a));
                            }
                        }
                        W[((size_t)mt * T.nt + tt) * WAVE + lane] = val;
                    }
            HIPCHK(hipMalloc((void **)&T.W, W.size() * sizeof(double)));
            HIPCHK(hipMemcpy(T.W, W.data(), W.size() * sizeof(double), hipMemcpyHostToDevice));
        }
        std::vector<double> tw((size_t)2 * H);
        for (int k = 0; k < H; ++k) {
            const double a = M_PI * (double)(2 * k + 1) / (double)L;
            tw[2 * (size_t)k] = cos(a);
            tw[2 * (size_t)k + 1] = -sin(a);
        }
        HIPCHK(hipMalloc((void **)&h->d_r2_tw, tw.size() * sizeof(double)));
        HIPCHK(hipMemcpy(h->d_r2_tw, tw.data(), tw.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    return ELPH_OK;
}

void elph_dft_mfma_free(elph_handle_s *h) {
    for (auto &a : h->mf) for (auto &T : a) if (T.W) { (void)hipFree(T.W); T.W = nullptr; }
    for (auto &T : h->mf_r2) if (T.W) { (void)hipFree(T.W); T.W = nullptr; }
    if (h->d_r2_tw) { (void)hipFree(h->d_r2_tw); h->d_r2_tw = nullptr; }
}
