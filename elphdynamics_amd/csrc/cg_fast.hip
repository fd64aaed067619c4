// cg_fast.hip — latency-tuned gfx950 kernels for lattices whose checkerboard has <= 4 colours
// (every even-L square and honeycomb lattice of the reference's example decks).
//
// What differs from the generic kernels in kernels.hip (same arithmetic, same results):
//   * "lane program": the bond list is re-packed on the host per colour into [colour][pass][lane]
//     so that every lane keeps ITS bonds (site pair + cosh + sinh) in registers for the whole kernel;
//     a checkerboard colour is then  LDS read -> 4 FMAs -> LDS write -> wave barrier  with no
//     dependent global load inside the sweep (the generic kernel pays one L2 round trip per colour);
//   * every global load of the kernel (state, partial sums, vectors, exp(-dtau V), lane program)
//     is independent of every other and issued up front: one memory round trip per kernel;
//   * the p ping-pong index is a launch constant (launch parity) instead of device state;
//   * XCD-aware 1-D grid: workgroups with equal (blockIdx.x % 8) share an XCD/L2, so they are given
//     consecutive tau-slices — the tau+-1 halo re-reads of the fused MtM then hit the same L2.
//
// Reference semantics: see kernels.hip / SURVEY.md Appendix A.

#include <cstdlib>

#include "elph_internal.h"

#define WAVE ELPH_WAVE

// Ordering of LDS traffic inside ONE wavefront.  Every slab in this file is private to a wave, and the LDS
// pipeline executes a wave's DS instructions in issue order, so a ds_read issued after a ds_write of the same
// wave observes it without any s_waitcnt/s_barrier in between.  All that is needed is that the COMPILER keeps
// the program order of possibly-aliasing LDS accesses: a pure compiler barrier, no instruction.
// (Using __syncthreads() here costs an s_waitcnt lgkmcnt(0) per colour: one extra LDS round trip per stage.)
#ifdef ELPH_LDS_SYNC
#define WAVE_LDS_ORDER() __syncthreads()
#else
#define WAVE_LDS_ORDER() asm volatile("" ::: "memory")
#endif

// device-coherent scalar traffic (experiment): agent-scope relaxed atomics => sc1 loads/stores that bypass L1/K$
__device__ __forceinline__ double ld_coh(const double *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ CgState ld_state(const CgState *p) {
    CgState s;
    const unsigned long long *q = reinterpret_cast<const unsigned long long *>(p);
    unsigned long long w[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) w[i] = __hip_atomic_load(q + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_memcpy(&s, w, sizeof(CgState));
    return s;
}

__device__ __forceinline__ double wave_sum2(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}

__device__ __forceinline__ double reduce_partials2(const double *p, int n) {
    double a = 0.0;
    for (int i = threadIdx.x; i < n; i += WAVE) a += ld_coh(p + i);
    return wave_sum2(a);
}

// XCD-aware mapping of a 1-D grid of 8*C*nrhs workgroups onto (tau, rhs); C = ceil(L/8)
__device__ __forceinline__ bool xcd_map(int L, int &t, int &rhs) {
    const int b = blockIdx.x;
    const int C = (L + 7) >> 3;
    const int xcd = b & 7, k = b >> 3;
    rhs = k / C;
    t = xcd * C + (k - rhs * C);
    return t < L;
}

template <int NPL>
struct LaneProg {
    static constexpr int PP = (NPL + 1) / 2;     // passes per colour: ceil(N/2 / 64)
    static constexpr int NE = 4 * PP;
    unsigned ij[NE];
};

template <int NPL>
__device__ __forceinline__ void lp_load_ij(unsigned (&ij)[4 * ((NPL + 1) / 2)], const ModelDev &m) {
    constexpr int NE = 4 * ((NPL + 1) / 2);
#pragma unroll
    for (int e = 0; e < NE; ++e) ij[e] = m.lp_ij[e * WAVE + threadIdx.x];
}

template <int NPL>
__device__ __forceinline__ void lp_load_cs(double (&c)[4 * ((NPL + 1) / 2)], double (&s)[4 * ((NPL + 1) / 2)],
                                           const double *lc, const double *ls) {
    constexpr int NE = 4 * ((NPL + 1) / 2);
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        c[e] = lc[e * WAVE + threadIdx.x];
        s[e] = ls[e * WAVE + threadIdx.x];
    }
}

// one checkerboard sweep with register-resident bonds (Checkerboard.jl:57-83 / :149-175).
// Straight-line code: idle lanes of a ragged colour address two private padding slots of the slab with
// (c,s) = (1,0), so there is no per-lane predicate and the compiler batches all LDS reads of a colour.
template <int NPL, int NBUF, bool REVERSE>
__device__ __forceinline__ void lp_sweep(double *buf0, double *buf1, const unsigned (&ij)[4 * ((NPL + 1) / 2)],
                                         const double (&c0)[4 * ((NPL + 1) / 2)], const double (&s0)[4 * ((NPL + 1) / 2)],
                                         const double (&c1)[4 * ((NPL + 1) / 2)], const double (&s1)[4 * ((NPL + 1) / 2)],
                                         int ncol) {
    constexpr int PP = (NPL + 1) / 2;
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
        const int col = REVERSE ? 3 - cc : cc;
        if (col < ncol) {
            double a0[PP], a1[PP], b0[PP], b1[PP];
#pragma unroll
            for (int pp = 0; pp < PP; ++pp) {
                const unsigned w = ij[col * PP + pp];
                const int i = w & 0xFFFF, j = w >> 16;
                a0[pp] = buf0[i]; a1[pp] = buf0[j];
                if (NBUF == 2) { b0[pp] = buf1[i]; b1[pp] = buf1[j]; }
            }
#pragma unroll
            for (int pp = 0; pp < PP; ++pp) {
                const int e = col * PP + pp;
                const unsigned w = ij[e];
                const int i = w & 0xFFFF, j = w >> 16;
                buf0[i] = c0[e] * a0[pp] + s0[e] * a1[pp];
                buf0[j] = c0[e] * a1[pp] + s0[e] * a0[pp];
                if (NBUF == 2) {
                    buf1[i] = c1[e] * b0[pp] + s1[e] * b1[pp];
                    buf1[j] = c1[e] * b1[pp] + s1[e] * b0[pp];
                }
            }
            WAVE_LDS_ORDER();
        }
    }
}

// LDS slab of one tau-slice: NPL*64 site slots + 2 private padding slots per lane (idle lane-program entries)
template <int NPL>
__host__ __device__ constexpr int slab_len() { return NPL * WAVE + 2 * WAVE; }

// ------------------------------------------------------------------------------------------
// y = M v | M^T v | M^T M v   (same maths as k_mul in kernels.hip)
// All global loads are unconditional (site index clamped to N-1 for the lanes past a ragged N); LDS
// accesses need no predicate (slab padded to NPL*64); only global stores are guarded.
// ------------------------------------------------------------------------------------------

template <int NPL, int WHICH, bool SSH>
__global__ void __launch_bounds__(WAVE) k_mul_fast(double *__restrict__ y, const double *__restrict__ v, ModelDev m) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int NE = 4 * ((NPL + 1) / 2);
    double *bufA = lds, *bufB = lds + slab_len<NPL>();
    const int N = m.N, L = m.L;
    int t, vecid;
    if (!xcd_map(L, t, vecid)) return;
    const int tm1 = (t == 0) ? L - 1 : t - 1;
    const int tp1 = (t == L - 1) ? 0 : t + 1;
    const size_t vec = (size_t)vecid * (size_t)N * (size_t)L;
    const double *vv = v + vec;
    double *yy = y + vec;
    const double sg0 = (t == 0) ? -1.0 : 1.0, sg1 = (tp1 == 0) ? -1.0 : 1.0;
    const double *Ech = m.E + (size_t)(vecid % m.nchains) * m.E_chain_stride;
    const double *E0 = Ech + (size_t)t * m.E_tau_stride, *E1 = Ech + (size_t)tp1 * m.E_tau_stride;

    unsigned ij[NE];
    double c0[NE], s0[NE], c1[NE], s1[NE];
    lp_load_ij<NPL>(ij, m);
    if (SSH) {
        lp_load_cs<NPL>(c0, s0, m.lp_c + (size_t)t * m.lp_tau_stride, m.lp_s + (size_t)t * m.lp_tau_stride);
        lp_load_cs<NPL>(c1, s1, m.lp_c + (size_t)tp1 * m.lp_tau_stride, m.lp_s + (size_t)tp1 * m.lp_tau_stride);
    } else {
        lp_load_cs<NPL>(c0, s0, m.lp_c, m.lp_s);
    }
    double vm[NPL], v0[NPL], vp[NPL], e0[NPL], e1[NPL], w0[NPL];
#pragma unroll
    for (int q = 0; q < NPL; ++q) {
        const int s = threadIdx.x + q * WAVE;
        const int sc = (s < N) ? s : N - 1;
        v0[q] = vv[(size_t)t * N + sc];
        if (WHICH != 1) { vm[q] = vv[(size_t)tm1 * N + sc]; e0[q] = E0[sc]; }
        if (WHICH != 0) { vp[q] = vv[(size_t)tp1 * N + sc]; e1[q] = E1[sc]; }
    }
    if (WHICH == 0) {
#pragma unroll
        for (int q = 0; q < NPL; ++q) bufA[threadIdx.x + q * WAVE] = e0[q] * vm[q];
        WAVE_LDS_ORDER();
        lp_sweep<NPL, 1, false>(bufA, nullptr, ij, c0, s0, c0, s0, m.ncol);
#pragma unroll
        for (int q = 0; q < NPL; ++q) { const int s = threadIdx.x + q * WAVE; if (s < N) yy[(size_t)t * N + s] = v0[q] - sg0 * bufA[s]; }
    } else if (WHICH == 1) {
#pragma unroll
        for (int q = 0; q < NPL; ++q) bufA[threadIdx.x + q * WAVE] = vp[q];
        WAVE_LDS_ORDER();
        if (SSH) lp_sweep<NPL, 1, true>(bufA, nullptr, ij, c1, s1, c1, s1, m.ncol);
        else lp_sweep<NPL, 1, true>(bufA, nullptr, ij, c0, s0, c0, s0, m.ncol);
#pragma unroll
        for (int q = 0; q < NPL; ++q) { const int s = threadIdx.x + q * WAVE; if (s < N) yy[(size_t)t * N + s] = v0[q] - sg1 * e1[q] * bufA[s]; }
    } else {
#pragma unroll
        for (int q = 0; q < NPL; ++q) {
            const int s = threadIdx.x + q * WAVE;
            bufA[s] = e0[q] * vm[q];
            bufB[s] = e1[q] * v0[q];
        }
        WAVE_LDS_ORDER();
        if (SSH) lp_sweep<NPL, 2, false>(bufA, bufB, ij, c0, s0, c1, s1, m.ncol);
        else lp_sweep<NPL, 2, false>(bufA, bufB, ij, c0, s0, c0, s0, m.ncol);
#pragma unroll
        for (int q = 0; q < NPL; ++q) {
            const int s = threadIdx.x + q * WAVE;
            w0[q] = v0[q] - sg0 * bufA[s];
            bufB[s] = vp[q] - sg1 * bufB[s];
        }
        WAVE_LDS_ORDER();
        if (SSH) lp_sweep<NPL, 1, true>(bufB, nullptr, ij, c1, s1, c1, s1, m.ncol);
        else lp_sweep<NPL, 1, true>(bufB, nullptr, ij, c0, s0, c0, s0, m.ncol);
#pragma unroll
        for (int q = 0; q < NPL; ++q) { const int s = threadIdx.x + q * WAVE; if (s < N) yy[(size_t)t * N + s] = w0[q] - sg1 * e1[q] * bufB[s]; }
    }
}

// ------------------------------------------------------------------------------------------
// CG kernels (IterativeSolvers.jl:153-314); see kernels.hip for the protocol between them.
// ------------------------------------------------------------------------------------------

#define IS_LEADER (t == 0)
template <int NPL, bool SSH>
__global__ void __launch_bounds__(WAVE) k_cg_ap_fast(CgBufs B, ModelDev m, int parity) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int NE = 4 * ((NPL + 1) / 2);
    double *bufA = lds, *bufB = lds + slab_len<NPL>();
    const int N = m.N, L = m.L;
    int t, rhs;
    if (!xcd_map(L, t, rhs)) return;
    const size_t ndim = (size_t)N * L;
    const int tm1 = (t == 0) ? L - 1 : t - 1;
    const int tp1 = (t == L - 1) ? 0 : t + 1;

    // ---- every global load of the kernel, none depending on another ---------------------------
    CgState *st2 = B.state + 2 * rhs;
    const CgState S = ld_state(st2 + parity);        // this launch's copy; the other one is written below
    const CgParams P = B.params;
    const double *src = (P.use_prec ? B.zp : B.r) + (size_t)rhs * ndim;
    const double *pold = B.p + ((size_t)parity * B.nrhs + rhs) * ndim;
    double *pnew = B.p + ((size_t)(parity ^ 1) * B.nrhs + rhs) * ndim;
    double *z = B.z + (size_t)rhs * ndim;
    const double *Ech = m.E + (size_t)(rhs % m.nchains) * m.E_chain_stride;
    const double *E0 = Ech + (size_t)t * m.E_tau_stride, *E1 = Ech + (size_t)tp1 * m.E_tau_stride;

    double *x = B.x + (size_t)rhs * ndim;
    const double alpha_prev = B.alpha[rhs];          // step length of the previous iteration (unused when seq == 0)
    double sm[NPL], s0v[NPL], sp[NPL], qm[NPL], q0[NPL], qp[NPL], e0[NPL], e1[NPL], xv[NPL];
#pragma unroll
    for (int q = 0; q < NPL; ++q) {
        const int s = threadIdx.x + q * WAVE;
        const int sc = (s < N) ? s : N - 1;
        const size_t im = (size_t)tm1 * N + sc, i0 = (size_t)t * N + sc, ip = (size_t)tp1 * N + sc;
        sm[q] = src[im]; s0v[q] = src[i0]; sp[q] = src[ip];
        qm[q] = pold[im]; q0[q] = pold[i0]; qp[q] = pold[ip];
        e0[q] = E0[sc]; e1[q] = E1[sc];
        xv[q] = x[i0];
    }
    unsigned ij[NE];
    double c0[NE], s0[NE], c1[NE], s1[NE];
    lp_load_ij<NPL>(ij, m);
    if (SSH) {
        lp_load_cs<NPL>(c0, s0, m.lp_c + (size_t)t * m.lp_tau_stride, m.lp_s + (size_t)t * m.lp_tau_stride);
        lp_load_cs<NPL>(c1, s1, m.lp_c + (size_t)tp1 * m.lp_tau_stride, m.lp_s + (size_t)tp1 * m.lp_tau_stride);
    } else {
        lp_load_cs<NPL>(c0, s0, m.lp_c, m.lp_s);
    }
    const double rr = reduce_partials2(B.rr + (size_t)rhs * L, L);
    const double rz = P.use_prec ? reduce_partials2(B.rz + (size_t)rhs * B.nrz, B.nrz) : rr;

    // ---- scalar control (identical in every wave of this rhs) ----------------------------------
    CgState *Sout = st2 + (parity ^ 1);
    if (S.done) {                                     // keep both copies terminal: later launches alternate between them
        if (IS_LEADER && threadIdx.x == 0) *Sout = S;
        return;
    }
    const long long seq = S.seq;
    const bool first = (seq == 0);
    double beta = 0.0, rho = S.rho, kmin = S.kmin, eps = S.eps;
    if (!first) {
        // x += alpha p of the iteration whose stop test follows (IterativeSolvers.jl:205/282), with the p this launch reads anyway
#pragma unroll
        for (int q = 0; q < NPL; ++q) {
            const int s = threadIdx.x + q * WAVE;
            if (s < N) x[(size_t)t * N + s] = xv[q] + alpha_prev * q0[q];
        }
        eps = sqrt(rr) / S.normb;
        const double qq = 2.0 * (double)seq / log(2.0 * S.eps0 / eps);
        const double val = qq * qq;
        kmin = (val > kmin) ? val : kmin;
        int done = 0;
        if (eps < P.tol) done = 1;
        else if (kmin > P.kmax) done = 2;
        else if (seq >= P.maxiter) done = 3;
        if (t == 0 && threadIdx.x == 0 && P.record_hist) B.hist[(size_t)rhs * P.hist_stride + seq] = eps;
        if (done) {
            if (t == 0 && threadIdx.x == 0) {
                CgState o = S;
                o.kmin = kmin; o.eps = eps; o.seq = seq + 1; o.iters = seq; o.done = done;
                *Sout = o;
            }
            return;
        }
        beta = rz / S.rho;
        rho = rz;
    }

    // ---- p = (z|r) + beta p on slices t-1, t, t+1; stage E.*p into LDS ---------------------------
    double p0[NPL], pp[NPL], w0[NPL];
#pragma unroll
    for (int q = 0; q < NPL; ++q) {
        const int s = threadIdx.x + q * WAVE;
        const double pm = first ? qm[q] : sm[q] + beta * qm[q];
        p0[q] = first ? q0[q] : s0v[q] + beta * q0[q];
        pp[q] = first ? qp[q] : sp[q] + beta * qp[q];
        bufA[s] = e0[q] * pm;
        bufB[s] = e1[q] * p0[q];
        if (s < N) pnew[(size_t)t * N + s] = p0[q];
    }
    WAVE_LDS_ORDER();
    const double sg0 = (t == 0) ? -1.0 : 1.0, sg1 = (tp1 == 0) ? -1.0 : 1.0;
    if (SSH) lp_sweep<NPL, 2, false>(bufA, bufB, ij, c0, s0, c1, s1, m.ncol);
    else lp_sweep<NPL, 2, false>(bufA, bufB, ij, c0, s0, c0, s0, m.ncol);
#pragma unroll
    for (int q = 0; q < NPL; ++q) {
        const int s = threadIdx.x + q * WAVE;
        w0[q] = p0[q] - sg0 * bufA[s];
        bufB[s] = pp[q] - sg1 * bufB[s];
    }
    WAVE_LDS_ORDER();
    if (SSH) lp_sweep<NPL, 1, true>(bufB, nullptr, ij, c1, s1, c1, s1, m.ncol);
    else lp_sweep<NPL, 1, true>(bufB, nullptr, ij, c0, s0, c0, s0, m.ncol);
    double acc = 0.0;
#pragma unroll
    for (int q = 0; q < NPL; ++q) {
        const int s = threadIdx.x + q * WAVE;
        const double zz = w0[q] - sg1 * e1[q] * bufB[s];
        if (s < N) {
            z[(size_t)t * N + s] = zz;
            acc += p0[q] * zz;
        }
    }
    acc = wave_sum2(acc);
    if (threadIdx.x == 0) {
        B.pap[(size_t)rhs * B.npap + t] = acc;
        if (t == 0) {
            CgState o = S;
            o.rho = rho; o.kmin = kmin; o.eps = eps; o.seq = seq + 1; o.iters = seq; o.done = 0;
            *Sout = o;
        }
    }
}

#undef IS_LEADER
#define IS_LEADER (ch == 0)
// combined pass: forward sweep on bufA (slice tau+1) and reverse sweep on bufB (w(tau)) in the same 4 stages
template <int NPL>
__device__ __forceinline__ void lp_sweep_fr(double *bufA, double *bufB, const unsigned (&ij)[4 * ((NPL + 1) / 2)],
                                            const double (&cA)[4 * ((NPL + 1) / 2)], const double (&sA)[4 * ((NPL + 1) / 2)],
                                            const double (&cB)[4 * ((NPL + 1) / 2)], const double (&sB)[4 * ((NPL + 1) / 2)],
                                            int ncol, bool doA) {
    constexpr int PP = (NPL + 1) / 2;
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
        const int colA = cc, colB = 3 - cc;
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
                    bufA[w & 0xFFFF] = cA[e] * a0[pp] + sA[e] * a1[pp];
                    bufA[w >> 16] = cA[e] * a1[pp] + sA[e] * a0[pp];
                }
                if (onB) {
                    const int e = colB * PP + pp; const unsigned w = ij[e];
                    bufB[w & 0xFFFF] = cB[e] * b0[pp] + sB[e] * b1[pp];
                    bufB[w >> 16] = cB[e] * b1[pp] + sB[e] * b0[pp];
                }
            }
            WAVE_LDS_ORDER();
        }
    }
}

// Batched-throughput variant of k_cg_ap_fast: one wave owns T consecutive tau-slices of one right-hand side.
//   w(t) = p(t) - sg(t) CB_t [E(t) .* p(t-1)]            needs T+1 forward sweeps  (t = t0 .. t0+T)
//   z(t) = w(t) - sg(t+1) E(t+1) .* CB_{t+1}^T w(t+1)    needs T   reverse sweeps
// The reverse sweep of w(t0+j) and the forward sweep of slice t0+j+1 run in the SAME four colour stages
// (independent LDS slabs), so a slice costs 4 stages instead of 8 and ~3 slice loads instead of 8; the next
// slice's loads are issued one stage ahead.  Results are bit-identical to the T=1 kernel (same operations per
// element, same order), only the p.z partial sums are grouped per chunk instead of per slice.
template <int NPL, int T, bool SSH>
__global__ void __launch_bounds__(WAVE) k_cg_ap_chunk(CgBufs B, ModelDev m, int parity) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int NE = 4 * ((NPL + 1) / 2);
    double *bufA = lds, *bufB = lds + slab_len<NPL>();
    const int N = m.N, L = m.L;
    const int nch = L / T;
    const int rhs = blockIdx.x / nch, ch = blockIdx.x - rhs * nch;
    const int t0 = ch * T;
    const size_t ndim = (size_t)N * L;
    auto wrap = [L](int t) { return (t < 0) ? t + L : ((t >= L) ? t - L : t); };

    CgState *st2 = B.state + 2 * rhs;
    const CgState S = ld_state(st2 + parity);        // this launch's copy; the other one is written below
    const CgParams P = B.params;
    const double *src = (P.use_prec ? B.zp : B.r) + (size_t)rhs * ndim;
    const double *pold = B.p + ((size_t)parity * B.nrhs + rhs) * ndim;
    double *pnew = B.p + ((size_t)(parity ^ 1) * B.nrhs + rhs) * ndim;
    double *z = B.z + (size_t)rhs * ndim;
    double *x = B.x + (size_t)rhs * ndim;
    const double alpha_prev = B.alpha[rhs];          // step length of the previous iteration (unused when seq == 0)

    int sc[NPL];
#pragma unroll
    for (int q = 0; q < NPL; ++q) { const int s = threadIdx.x + q * WAVE; sc[q] = (s < N) ? s : N - 1; }
    auto load_sq = [&](int t, double (&sv)[NPL], double (&qv)[NPL]) {
#pragma unroll
        for (int q = 0; q < NPL; ++q) { const size_t i = (size_t)t * N + sc[q]; sv[q] = src[i]; qv[q] = pold[i]; }
    };
    auto load_e = [&](int t, double (&ev)[NPL]) {
        const double *Et = m.E + (size_t)(rhs % m.nchains) * m.E_chain_stride + (size_t)t * m.E_tau_stride;
#pragma unroll
        for (int q = 0; q < NPL; ++q) ev[q] = Et[sc[q]];
    };
    auto load_x = [&](int t, double (&xv)[NPL]) {
#pragma unroll
        for (int q = 0; q < NPL; ++q) xv[q] = x[(size_t)t * N + sc[q]];
    };
    // x(t) += alpha_prev p_old(t) for an OWN slice (t0 <= t < t0+T), with the p_old values this launch has in registers
    auto update_x = [&](int t, const double (&xv)[NPL], const double (&qv)[NPL]) {
#pragma unroll
        for (int q = 0; q < NPL; ++q) {
            const int s = threadIdx.x + q * WAVE;
            if (s < N) x[(size_t)t * N + s] = xv[q] + alpha_prev * qv[q];
        }
    };

    // ---- prologue loads (independent of everything) ----------------------------------------------
    double Sm[NPL], Qm[NPL], S0[NPL], Q0[NPL], S1[NPL], Q1[NPL], E0[NPL], E1[NPL], X0[NPL], X1[NPL];
    load_sq(wrap(t0 - 1), Sm, Qm);
    load_sq(t0, S0, Q0);
    load_sq(wrap(t0 + 1), S1, Q1);
    load_e(t0, E0);
    load_e(wrap(t0 + 1), E1);
    load_x(t0, X0);
    if (T > 1) load_x(t0 + 1, X1);                   // t0 + 1 <= L - 1 whenever T > 1 (T divides L)
    unsigned ij[NE];
    double cA[NE], sA[NE], cB[NE], sB[NE];
    lp_load_ij<NPL>(ij, m);
    if (SSH) {
        lp_load_cs<NPL>(cA, sA, m.lp_c + (size_t)t0 * m.lp_tau_stride, m.lp_s + (size_t)t0 * m.lp_tau_stride);
        lp_load_cs<NPL>(cB, sB, m.lp_c + (size_t)wrap(t0 + 1) * m.lp_tau_stride, m.lp_s + (size_t)wrap(t0 + 1) * m.lp_tau_stride);
    } else {
        lp_load_cs<NPL>(cA, sA, m.lp_c, m.lp_s);
    }
    const double rr = reduce_partials2(B.rr + (size_t)rhs * L, L);
    const double rz = P.use_prec ? reduce_partials2(B.rz + (size_t)rhs * B.nrz, B.nrz) : rr;

    // ---- scalar control (identical in every wave of this rhs; same code as k_cg_ap_fast) -----------
    CgState *Sout = st2 + (parity ^ 1);
    if (S.done) {                                     // keep both copies terminal: later launches alternate between them
        if (IS_LEADER && threadIdx.x == 0) *Sout = S;
        return;
    }
    const long long seq = S.seq;
    const bool first = (seq == 0);
    double beta = 0.0, rho = S.rho, kmin = S.kmin, eps = S.eps;
    if (!first) {
        eps = sqrt(rr) / S.normb;
        const double qq = 2.0 * (double)seq / log(2.0 * S.eps0 / eps);
        const double val = qq * qq;
        kmin = (val > kmin) ? val : kmin;
        int done = 0;
        if (eps < P.tol) done = 1;
        else if (kmin > P.kmax) done = 2;
        else if (seq >= P.maxiter) done = 3;
        if (ch == 0 && threadIdx.x == 0 && P.record_hist) B.hist[(size_t)rhs * P.hist_stride + seq] = eps;
        if (done) {
            // the solve ends here: apply the pending x += alpha p of the last iteration to all own slices, then leave
            update_x(t0, X0, Q0);
            if (T > 1) update_x(t0 + 1, X1, Q1);
            for (int j = 2; j < T; ++j) {
                double xv[NPL], qv[NPL];
#pragma unroll
                for (int q = 0; q < NPL; ++q) { const size_t i = (size_t)(t0 + j) * N + sc[q]; xv[q] = x[i]; qv[q] = pold[i]; }
                update_x(t0 + j, xv, qv);
            }
            if (ch == 0 && threadIdx.x == 0) {
                CgState o = S;
                o.kmin = kmin; o.eps = eps; o.seq = seq + 1; o.iters = seq; o.done = done;
                *Sout = o;
            }
            return;
        }
        update_x(t0, X0, Q0);
        if (T > 1) update_x(t0 + 1, X1, Q1);
        beta = rz / S.rho;
        rho = rz;
    }
    auto pval = [&](double sv, double qv) { return first ? qv : sv + beta * qv; };
    auto sgn = [](int t) { return (t == 0) ? -1.0 : 1.0; };

    // ---- prologue: w(t0), w(t0+1) by one two-slab forward sweep --------------------------------------
    double pprev[NPL], pcur[NPL], wprev[NPL], wcur[NPL], Ecur[NPL];
#pragma unroll
    for (int q = 0; q < NPL; ++q) {
        const int s = threadIdx.x + q * WAVE;
        const double pm = pval(Sm[q], Qm[q]);
        pprev[q] = pval(S0[q], Q0[q]);
        pcur[q] = pval(S1[q], Q1[q]);
        bufA[s] = E0[q] * pm;
        bufB[s] = E1[q] * pprev[q];
        Ecur[q] = E1[q];
        if (s < N) pnew[(size_t)t0 * N + s] = pprev[q];
    }
    WAVE_LDS_ORDER();
    if (SSH) lp_sweep<NPL, 2, false>(bufA, bufB, ij, cA, sA, cB, sB, m.ncol);
    else lp_sweep<NPL, 2, false>(bufA, bufB, ij, cA, sA, cA, sA, m.ncol);
    {
        const double sga = sgn(t0), sgb = sgn(wrap(t0 + 1));
#pragma unroll
        for (int q = 0; q < NPL; ++q) {
            const int s = threadIdx.x + q * WAVE;
            wprev[q] = pprev[q] - sga * bufA[s];
            wcur[q] = pcur[q] - sgb * bufB[s];
        }
    }
    WAVE_LDS_ORDER();

    // ---- pipelined stages: reverse sweep of w(t0+j)  ||  forward sweep of slice t0+j+1 ---------------
    double acc = 0.0;
    // Slices t0+2 … t0+T stream through a register ring PF stages deep: a stage (four colour sweeps) lasts ~0.3 us, an HBM
    // round trip under load several times that, and at ~1.25 waves per SIMD nothing else hides it — the registers are free.
    constexpr int PF = (T >= 4) ? 2 : 1;
    double Sr[PF][NPL], Qr[PF][NPL], Er[PF][NPL], Xr[PF][NPL];
#pragma unroll
    for (int k = 0; k < PF; ++k)
        if (k + 2 <= T) {
            load_sq(wrap(t0 + 2 + k), Sr[k], Qr[k]);
            load_e(wrap(t0 + 2 + k), Er[k]);
            if (k + 2 < T) load_x(t0 + 2 + k, Xr[k]);          // own slices only (the last ring slice, t0+T, is halo)
        }
#pragma unroll
    for (int j = 1; j <= T; ++j) {
        const int tj = wrap(t0 + j);              // slice whose w is reverse-swept now
        const int tn = wrap(t0 + j + 1);          // slice forward-swept now (if j < T)
        const bool more = (j < T);
        double (&Sn)[NPL] = Sr[(j - 1) % PF];     // slice tn in the ring (compile-time slot: the loop is unrolled)
        double (&Qn)[NPL] = Qr[(j - 1) % PF];
        double (&En)[NPL] = Er[(j - 1) % PF];
        double (&Xn)[NPL] = Xr[(j - 1) % PF];
        if (SSH) {
            // B-side tables: slice tj (they were the A/B tables of the previous stage); A-side: slice tn
            lp_load_cs<NPL>(cB, sB, m.lp_c + (size_t)tj * m.lp_tau_stride, m.lp_s + (size_t)tj * m.lp_tau_stride);
            if (more) lp_load_cs<NPL>(cA, sA, m.lp_c + (size_t)tn * m.lp_tau_stride, m.lp_s + (size_t)tn * m.lp_tau_stride);
        }
#pragma unroll
        for (int q = 0; q < NPL; ++q) {
            const int s = threadIdx.x + q * WAVE;
            bufB[s] = wcur[q];
            if (more) bufA[s] = En[q] * pcur[q];          // E(tn) .* p(tj)
        }
        WAVE_LDS_ORDER();
        if (SSH) lp_sweep_fr<NPL>(bufA, bufB, ij, cA, sA, cB, sB, m.ncol, more);
        else lp_sweep_fr<NPL>(bufA, bufB, ij, cA, sA, cA, sA, m.ncol, more);
        const double sgj = sgn(tj), sgnn = sgn(tn);
        double pnext[NPL], wnext[NPL];
#pragma unroll
        for (int q = 0; q < NPL; ++q) {
            const int s = threadIdx.x + q * WAVE;
            const double zz = wprev[q] - sgj * Ecur[q] * bufB[s];        // z(t0+j-1)
            if (s < N) {
                z[(size_t)wrap(t0 + j - 1) * N + s] = zz;
                acc += pprev[q] * zz;
            }
            if (more) {
                pnext[q] = pval(Sn[q], Qn[q]);                           // p(tn)
                wnext[q] = pnext[q] - sgnn * bufA[s];                    // w(tn)
                if (s < N) pnew[(size_t)tj * N + s] = pcur[q];
            }
        }
        if (!first && j + 1 < T) update_x(t0 + j + 1, Xn, Qn);           // tn = t0+j+1 is an own slice
        WAVE_LDS_ORDER();
        if (more) {
#pragma unroll
            for (int q = 0; q < NPL; ++q) {
                wprev[q] = wcur[q]; wcur[q] = wnext[q];
                pprev[q] = pcur[q]; pcur[q] = pnext[q];
                Ecur[q] = En[q];
            }
            if (j + 1 + PF <= T) {                                        // refill this slot
                load_sq(wrap(t0 + j + 1 + PF), Sn, Qn);
                load_e(wrap(t0 + j + 1 + PF), En);
                if (j + 1 + PF < T) load_x(t0 + j + 1 + PF, Xn);
            }
        }
    }
    acc = wave_sum2(acc);
    if (threadIdx.x == 0) {
        B.pap[(size_t)rhs * B.npap + ch] = acc;
        if (ch == 0) {
            CgState o = S;
            o.rho = rho; o.kmin = kmin; o.eps = eps; o.seq = seq + 1; o.iters = seq; o.done = 0;
            *Sout = o;
        }
    }
}

#undef IS_LEADER
template <int NPL>
__global__ void __launch_bounds__(WAVE) k_cg_xr_fast(CgBufs B, int N, int L, int parity) {
    // r -= alpha z and the partial r.r.  x += alpha p is NOT done here: the next k_cg_ap reads this p anyway (as its p_old),
    // so it applies the update there and this kernel moves 24 B per element instead of 48 (alpha travels in B.alpha).
    int t, rhs;
    if (!xcd_map(L, t, rhs)) return;
    const size_t ndim = (size_t)N * L;
    const CgState S = ld_state(B.state + 2 * rhs + parity);   // written by the k_cg_ap launch just before
    const double *z = B.z + (size_t)rhs * ndim;
    double *r = B.r + (size_t)rhs * ndim;
    double rv[NPL], zv[NPL];
#pragma unroll
    for (int q = 0; q < NPL; ++q) {
        const int s = threadIdx.x + q * WAVE;
        const size_t i = (size_t)t * N + ((s < N) ? s : N - 1);
        rv[q] = r[i]; zv[q] = z[i];
    }
    const double pap = reduce_partials2(B.pap + (size_t)rhs * B.npap, B.npap);
    if (S.done) return;
    const double alpha = S.rho / pap;
    double acc = 0.0;
#pragma unroll
    for (int q = 0; q < NPL; ++q) {
        const int s = threadIdx.x + q * WAVE;
        if (s < N) {
            const size_t i = (size_t)t * N + s;
            const double rn = rv[q] - alpha * zv[q];
            r[i] = rn;
            acc += rn * rn;
        }
    }
    acc = wave_sum2(acc);
    if (threadIdx.x == 0) {
        B.rr[(size_t)rhs * L + t] = acc;
        if (t == 0) B.alpha[rhs] = alpha;
    }
}

// ------------------------------------------------------------------------------------------
// KPM per-omega Chebyshev recursion with register-resident bonds (KPMPreconditioners.jl:606-693,758-778)
// ------------------------------------------------------------------------------------------

template <int NPL, bool REVERSE>
__device__ __forceinline__ void lp_sweep_z(double2 *buf, const unsigned (&ij)[4 * ((NPL + 1) / 2)],
                                           const double (&c)[4 * ((NPL + 1) / 2)], const double (&s)[4 * ((NPL + 1) / 2)],
                                           int ncol) {
    constexpr int PP = (NPL + 1) / 2;
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
        const int col = REVERSE ? 3 - cc : cc;
        if (col < ncol) {
            double2 a0[PP], a1[PP];
#pragma unroll
            for (int pp = 0; pp < PP; ++pp) {
                const unsigned w = ij[col * PP + pp];
                a0[pp] = buf[w & 0xFFFF]; a1[pp] = buf[w >> 16];
            }
#pragma unroll
            for (int pp = 0; pp < PP; ++pp) {
                const int e = col * PP + pp;
                const unsigned w = ij[e];
                buf[w & 0xFFFF] = make_double2(c[e] * a0[pp].x + s[e] * a1[pp].x, c[e] * a0[pp].y + s[e] * a1[pp].y);
                buf[w >> 16] = make_double2(c[e] * a1[pp].x + s[e] * a0[pp].x, c[e] * a1[pp].y + s[e] * a0[pp].y);
            }
            WAVE_LDS_ORDER();
        }
    }
}

template <int NPL, bool TRANSPOSED, bool CONJ>
__device__ __forceinline__ void kpm_series_fast(double2 (&acc)[NPL], const double2 (&vin)[NPL], double2 *buf,
                                                const double (&eb)[NPL], const double2 *c, int order, double a, double b,
                                                const unsigned (&ij)[4 * ((NPL + 1) / 2)], const double (&cb)[4 * ((NPL + 1) / 2)],
                                                const double (&sb)[4 * ((NPL + 1) / 2)], int ncol, int N) {
    double2 um1[NPL], un[NPL], up1[NPL];
    double2 c0 = c[0];
    if (CONJ) c0.y = -c0.y;
#pragma unroll
    for (int q = 0; q < NPL; ++q) {
        acc[q] = make_double2(c0.x * vin[q].x - c0.y * vin[q].y, c0.x * vin[q].y + c0.y * vin[q].x);
        un[q] = vin[q];
        um1[q] = make_double2(0.0, 0.0);
    }
    for (int n = 2; n <= order; ++n) {
        // up1 = A' un   (mulA'!, :685-693; A = CBbar diag(Ebar), A^T = diag(Ebar) CBbar^T, :758-778)
#pragma unroll
        for (int q = 0; q < NPL; ++q)
            buf[threadIdx.x + q * WAVE] = TRANSPOSED ? un[q] : make_double2(eb[q] * un[q].x, eb[q] * un[q].y);
        WAVE_LDS_ORDER();
        lp_sweep_z<NPL, TRANSPOSED>(buf, ij, cb, sb, ncol);
        const double2 cn0 = c[n - 1];
        const double2 cn = make_double2(cn0.x, CONJ ? -cn0.y : cn0.y);
#pragma unroll
        for (int q = 0; q < NPL; ++q) {
            double2 av = buf[threadIdx.x + q * WAVE];
            if (TRANSPOSED) { av.x *= eb[q]; av.y *= eb[q]; }
            up1[q] = make_double2(a * av.x - b * un[q].x, a * av.y - b * un[q].y);
            if (n > 2) {     // u_{n+1} = 2 A' u_n - u_{n-1}; the first step is u_2 = A' u_1
                up1[q].x = 2.0 * up1[q].x - um1[q].x;
                up1[q].y = 2.0 * up1[q].y - um1[q].y;
            }
            um1[q] = un[q];
            un[q] = up1[q];
            acc[q].x += cn.x * un[q].x - cn.y * un[q].y;
            acc[q].y += cn.x * un[q].y + cn.y * un[q].x;
        }
        WAVE_LDS_ORDER();
    }
}

template <int NPL>
__global__ void __launch_bounds__(WAVE) k_kpm_cheb_fast(double2 *__restrict__ nu, KpmDev K, ModelDev m, int Lo2,
                                                        const CgState *state) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int NE = 4 * ((NPL + 1) / 2);
    double2 *buf = reinterpret_cast<double2 *>(lds);
    const int rhs = blockIdx.x;   // x = right-hand side, y = frequency in longest-first order: ALL long recursions are dispatched first
    if (state && ld_state(state + 2 * rhs).done) return;   // `state` points at the current copy (host adds the parity)
    const KpmChainView V = kpm_chain_view(K, rhs, m.N);
    const int w = V.wsched[blockIdx.y];
    const int N = m.N;
    const int order = V.order[w];
    const double2 *c = K.coeff + V.coff[w];
    double2 *u = nu + ((size_t)rhs * Lo2 + w) * N;
    unsigned ij[NE];
    double cb[NE], sb[NE];
    lp_load_ij<NPL>(ij, m);
    lp_load_cs<NPL>(cb, sb, K.lp_cbar, K.lp_sbar);
    double2 vin[NPL], mid[NPL], res[NPL];
    double eb[NPL];
#pragma unroll
    for (int q = 0; q < NPL; ++q) {
        const int s = threadIdx.x + q * WAVE;
        const int sc = (s < N) ? s : N - 1;
        vin[q] = u[sc];
        eb[q] = V.Ebar[sc];
    }
    const double a = V.a, b = V.b;
    kpm_series_fast<NPL, true, true>(mid, vin, buf, eb, c, order, a, b, ij, cb, sb, m.ncol, N);
    kpm_series_fast<NPL, false, false>(res, mid, buf, eb, c, order, a, b, ij, cb, sb, m.ncol, N);
#pragma unroll
    for (int q = 0; q < NPL; ++q) {
        const int s = threadIdx.x + q * WAVE;
        if (s < N) u[s] = res[q];
    }
}

// ------------------------------------------------------------------------------------------
// KPM Chebyshev recursion, re/im-split variant.
// A' is real, so the real and the imaginary part of u_n obey the SAME real three-term recursion and never
// mix; only the coefficient sums do.  One 128-thread workgroup per frequency block: wave 0 carries Re u,
// wave 1 carries Im u, each in its own LDS slab (half the LDS bytes and half the instructions per wave of the
// complex kernel), each accumulating P = sum cx_n u_n and Q = sum cy_n u_n; the halves are combined
// through LDS once per series:
//     conj coefficients (first series):  Re = P_re + Q_im,  Im = P_im - Q_re
//     plain coefficients (second):       Re = P_re - Q_im,  Im = P_im + Q_re
// LDS addresses of a lane's bonds are precomputed pointers (no per-access address arithmetic).
// ------------------------------------------------------------------------------------------

template <int NPL, bool REVERSE>
__device__ __forceinline__ void lp_sweep_ptr(double *const (&pi)[4 * ((NPL + 1) / 2)], double *const (&pj)[4 * ((NPL + 1) / 2)],
                                             const double (&c)[4 * ((NPL + 1) / 2)], const double (&s)[4 * ((NPL + 1) / 2)],
                                             int ncol) {
    constexpr int PP = (NPL + 1) / 2;
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
        const int col = REVERSE ? 3 - cc : cc;
        if (col < ncol) {
            double a0[PP], a1[PP];
#pragma unroll
            for (int pp = 0; pp < PP; ++pp) { a0[pp] = *pi[col * PP + pp]; a1[pp] = *pj[col * PP + pp]; }
#pragma unroll
            for (int pp = 0; pp < PP; ++pp) {
                const int e = col * PP + pp;
                *pi[e] = c[e] * a0[pp] + s[e] * a1[pp];
                *pj[e] = c[e] * a1[pp] + s[e] * a0[pp];
            }
            WAVE_LDS_ORDER();
        }
    }
}

template <int NPL, bool TRANSPOSED>
__device__ __forceinline__ void kpm_series_ri(double (&P)[NPL], double (&Q)[NPL], const double (&vin)[NPL], double *slab,
                                              const double (&eb)[NPL], const double2 *c, int order, double a, double b,
                                              double *const (&pi)[4 * ((NPL + 1) / 2)], double *const (&pj)[4 * ((NPL + 1) / 2)],
                                              const double (&cb)[4 * ((NPL + 1) / 2)], const double (&sb)[4 * ((NPL + 1) / 2)],
                                              int ncol) {
    double um1[NPL], un[NPL];
    const int lane = threadIdx.x & (WAVE - 1);
    {
        const double2 c0 = c[0];
#pragma unroll
        for (int q = 0; q < NPL; ++q) {
            P[q] = c0.x * vin[q];
            Q[q] = c0.y * vin[q];
            un[q] = vin[q];
            um1[q] = 0.0;
        }
    }
    for (int n = 2; n <= order; ++n) {
#pragma unroll
        for (int q = 0; q < NPL; ++q) slab[lane + q * WAVE] = TRANSPOSED ? un[q] : eb[q] * un[q];
        WAVE_LDS_ORDER();
        lp_sweep_ptr<NPL, TRANSPOSED>(pi, pj, cb, sb, ncol);
        const double2 cn = c[n - 1];
#pragma unroll
        for (int q = 0; q < NPL; ++q) {
            double av = slab[lane + q * WAVE];
            if (TRANSPOSED) av *= eb[q];
            double up = a * av - b * un[q];                    // A' u_n   (mulA'!, :685-693)
            if (n > 2) up = 2.0 * up - um1[q];                 // u_{n+1} = 2 A' u_n - u_{n-1}
            um1[q] = un[q];
            un[q] = up;
            P[q] += cn.x * up;
            Q[q] += cn.y * up;
        }
        WAVE_LDS_ORDER();
    }
}

template <int NPL>
__global__ void __launch_bounds__(2 * WAVE) k_kpm_cheb_ri(double2 *__restrict__ nu, KpmDev K, ModelDev m, int Lo2,
                                                          const CgState *state) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int NE = 4 * ((NPL + 1) / 2);
    constexpr int SL = slab_len<NPL>();
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & (WAVE - 1);
    double *slab = lds + wv * SL;                 // this wave's component slab
    double *xch = lds + 2 * SL;                   // exchange area [2][NPL*64]
    const int rhs = blockIdx.x;   // x = right-hand side, y = frequency in longest-first order: ALL long recursions are dispatched first
    if (state && ld_state(state + 2 * rhs).done) return;   // `state` points at the current copy (host adds the parity)
    const KpmChainView V = kpm_chain_view(K, rhs, m.N);
    const int w = V.wsched[blockIdx.y];
    const int N = m.N;
    const int order = V.order[w];
    const double2 *c = K.coeff + V.coff[w];
    double *u = reinterpret_cast<double *>(nu + ((size_t)rhs * Lo2 + w) * N);    // interleaved re,im
    double *pi[NE], *pj[NE];
    double cb[NE], sb[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const unsigned ij = m.lp_ij[e * WAVE + lane];
        pi[e] = slab + (ij & 0xFFFF);
        pj[e] = slab + (ij >> 16);
        cb[e] = K.lp_cbar[e * WAVE + lane];
        sb[e] = K.lp_sbar[e * WAVE + lane];
    }
    double vin[NPL], eb[NPL], P[NPL], Q[NPL], mid[NPL];
#pragma unroll
    for (int q = 0; q < NPL; ++q) {
        const int s = lane + q * WAVE;
        const int sc = (s < N) ? s : N - 1;
        vin[q] = u[2 * sc + wv];
        eb[q] = V.Ebar[sc];
    }
    const double a = V.a, b = V.b;
    // ---- first series: M^-T[w,w], conjugated coefficients (KPMPreconditioners.jl:621-648)
    kpm_series_ri<NPL, true>(P, Q, vin, slab, eb, c, order, a, b, pi, pj, cb, sb, m.ncol);
#pragma unroll
    for (int q = 0; q < NPL; ++q) xch[wv * NPL * WAVE + lane + q * WAVE] = Q[q];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NPL; ++q) {
        const double Qo = xch[(wv ^ 1) * NPL * WAVE + lane + q * WAVE];
        mid[q] = (wv == 0) ? P[q] + Qo : P[q] - Qo;
    }
    __syncthreads();
    // ---- second series: M^-1[w,w] (:650-677)
    kpm_series_ri<NPL, false>(P, Q, mid, slab, eb, c, order, a, b, pi, pj, cb, sb, m.ncol);
#pragma unroll
    for (int q = 0; q < NPL; ++q) xch[wv * NPL * WAVE + lane + q * WAVE] = Q[q];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NPL; ++q) {
        const int s = lane + q * WAVE;
        const double Qo = xch[(wv ^ 1) * NPL * WAVE + lane + q * WAVE];
        const double res = (wv == 0) ? P[q] - Qo : P[q] + Qo;
        if (s < N) u[2 * s + wv] = res;
    }
}

// ------------------------------------------------------------------------------------------
// KPM Chebyshev recursion for the even-L square lattice (L = 8, 16): checkerboard exchange in REGISTERS.
// The greedy colouring of the reference yields [x-even | x-odd | y-even | y-odd] (the
// host verifies the bond table against exactly that pattern before enabling this kernel).  Each lane owns a
// P x P patch of sites (P = L/8; lanes form an 8 x 8 grid of patches):
//   * x-even / y-even bonds of a 2x2 patch connect two of the lane's own registers  -> no data movement,
//   * x-odd / y-odd bonds connect to the neighbouring patch                          -> one wave shuffle per value,
// so a checkerboard apply is 2 shuffle rounds + FMAs, with no LDS slab, no LDS round trip per colour.
// Same re/im split as k_kpm_cheb_ri (wave 0 = Re, wave 1 = Im, one LDS exchange per series).
// ------------------------------------------------------------------------------------------

template <int P>
struct SqLane {
    static constexpr int NS = P * P;
    double c[4][P * P], s[4][P * P];     // per colour, per own site: cosh/sinh of the bond touching it
    int xp, xm, yp, ym, xe, ye;          // partner lanes: +x, -x, +y, -y neighbours; P == 1: x-even / y-even partner
};

template <int P, bool REVERSE>
__device__ __forceinline__ void sq_cb_apply(double (&v)[P * P], const SqLane<P> &T) {
    // slot index: dx + P*dy
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
        const int col = REVERSE ? 3 - cc : cc;
        if (P == 2) {
            if (col == 0 || col == 2) {          // in-lane pairs: (0,d)-(1,d) along x, (d,0)-(d,1) along y
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    const int i = (col == 0) ? (0 + 2 * d) : (d + 0), j = (col == 0) ? (1 + 2 * d) : (d + 2);
                    const double t0 = v[i], t1 = v[j];
                    v[i] = T.c[col][i] * t0 + T.s[col][i] * t1;
                    v[j] = T.c[col][j] * t1 + T.s[col][j] * t0;
                }
            } else {                             // cross-lane: my high-side sites pair with the +neighbour's low-side sites
                const int up = (col == 1) ? T.xp : T.yp, dn = (col == 1) ? T.xm : T.ym;
                double fromUp[2], fromDn[2];
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    const int lo = (col == 1) ? (0 + 2 * d) : (d + 0), hi = (col == 1) ? (1 + 2 * d) : (d + 2);
                    fromUp[d] = __shfl(v[lo], up, WAVE);      // neighbour's low-side value -> partner of my high-side site
                    fromDn[d] = __shfl(v[hi], dn, WAVE);      // neighbour's high-side value -> partner of my low-side site
                }
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    const int lo = (col == 1) ? (0 + 2 * d) : (d + 0), hi = (col == 1) ? (1 + 2 * d) : (d + 2);
                    v[hi] = T.c[col][hi] * v[hi] + T.s[col][hi] * fromUp[d];
                    v[lo] = T.c[col][lo] * v[lo] + T.s[col][lo] * fromDn[d];
                }
            }
        } else {                                 // P == 1: one site per lane, every colour is a lane permutation
            const int partner = (col == 0) ? T.xe : (col == 2) ? T.ye : (col == 1) ? T.xp : T.yp;   // xp/yp hold the odd-colour partner
            const double t = __shfl(v[0], partner, WAVE);
            v[0] = T.c[col][0] * v[0] + T.s[col][0] * t;
        }
    }
}

template <int P, bool TRANSPOSED>
__device__ __forceinline__ void kpm_series_sq(double (&Pacc)[P * P], double (&Qacc)[P * P], const double (&vin)[P * P],
                                              const double (&eb)[P * P], const double2 *c, int order, double a, double b,
                                              const SqLane<P> &T) {
    constexpr int NS = P * P;
    double um1[NS], un[NS];
    {
        const double2 c0 = c[0];
#pragma unroll
        for (int q = 0; q < NS; ++q) { Pacc[q] = c0.x * vin[q]; Qacc[q] = c0.y * vin[q]; un[q] = vin[q]; um1[q] = 0.0; }
    }
    for (int n = 2; n <= order; ++n) {
        double w[NS];
#pragma unroll
        for (int q = 0; q < NS; ++q) w[q] = TRANSPOSED ? un[q] : eb[q] * un[q];
        sq_cb_apply<P, TRANSPOSED>(w, T);
        const double2 cn = c[n - 1];
#pragma unroll
        for (int q = 0; q < NS; ++q) {
            double av = w[q];
            if (TRANSPOSED) av *= eb[q];
            double up = a * av - b * un[q];
            if (n > 2) up = 2.0 * up - um1[q];
            um1[q] = un[q];
            un[q] = up;
            Pacc[q] += cn.x * up;
            Qacc[q] += cn.y * up;
        }
    }
}

template <int P>
__global__ void __launch_bounds__(2 * WAVE) k_kpm_cheb_sq(double2 *__restrict__ nu, KpmDev K, const double *__restrict__ sqc,
                                                          const double *__restrict__ sqs, int N, int Lo2,
                                                          const CgState *state) {
    constexpr int NS = P * P, LS = 8 * P;
    __shared__ double xch[2][NS * WAVE];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & (WAVE - 1);
    const int rhs = blockIdx.x;   // x = right-hand side, y = frequency in longest-first order: ALL long recursions are dispatched first
    if (state && ld_state(state + 2 * rhs).done) return;
    const KpmChainView V = kpm_chain_view(K, rhs, N);
    const int w = V.wsched[blockIdx.y];
    const int order = V.order[w];
    const double2 *c = K.coeff + V.coff[w];
    double *u = reinterpret_cast<double *>(nu + ((size_t)rhs * Lo2 + w) * N);
    const int pa = lane & 7, pb = lane >> 3;
    SqLane<P> T;
    T.xp = ((pa + 1) & 7) + 8 * pb; T.xm = ((pa + 7) & 7) + 8 * pb;
    T.yp = pa + 8 * ((pb + 1) & 7); T.ym = pa + 8 * ((pb + 7) & 7);
    T.xe = (pa ^ 1) + 8 * pb; T.ye = pa + 8 * (pb ^ 1);
    if (P == 1) {   // odd colours pair (odd, odd+1): an odd coordinate looks up, an even one looks down
        if (!(pa & 1)) T.xp = T.xm;
        if (!(pb & 1)) T.yp = T.ym;
    }
    int site[NS];
    double vin[NS], eb[NS], Pa[NS], Qa[NS], mid[NS];
#pragma unroll
    for (int q = 0; q < NS; ++q) {
        const int dx = q % P, dy = q / P;
        site[q] = (pa * P + dx) + LS * (pb * P + dy);
        vin[q] = u[2 * site[q] + wv];
        eb[q] = V.Ebar[site[q]];
#pragma unroll
        for (int col = 0; col < 4; ++col) {
            T.c[col][q] = sqc[(size_t)col * N + site[q]];
            T.s[col][q] = sqs[(size_t)col * N + site[q]];
        }
    }
    const double a = V.a, b = V.b;
    kpm_series_sq<P, true>(Pa, Qa, vin, eb, c, order, a, b, T);
#pragma unroll
    for (int q = 0; q < NS; ++q) xch[wv][q * WAVE + lane] = Qa[q];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NS; ++q) {
        const double Qo = xch[wv ^ 1][q * WAVE + lane];
        mid[q] = (wv == 0) ? Pa[q] + Qo : Pa[q] - Qo;
    }
    __syncthreads();
    kpm_series_sq<P, false>(Pa, Qa, mid, eb, c, order, a, b, T);
#pragma unroll
    for (int q = 0; q < NS; ++q) xch[wv][q * WAVE + lane] = Qa[q];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NS; ++q) {
        const double Qo = xch[wv ^ 1][q * WAVE + lane];
        u[2 * site[q] + wv] = (wv == 0) ? Pa[q] - Qo : Pa[q] + Qo;
    }
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------

#define DISPATCH_NPL_F(npl, CALL)                                 \
    switch (npl) {                                                \
        case 1: { constexpr int NPL = 1; CALL; } break;           \
        case 2: { constexpr int NPL = 2; CALL; } break;           \
        case 3: { constexpr int NPL = 3; CALL; } break;           \
        case 4: { constexpr int NPL = 4; CALL; } break;           \
        case 5: { constexpr int NPL = 5; CALL; } break;           \
        case 6: { constexpr int NPL = 6; CALL; } break;           \
        case 7: { constexpr int NPL = 7; CALL; } break;           \
        default: { constexpr int NPL = 8; CALL; } break;          \
    }

static int check_launch_f(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        elph_set_error("launch %s failed: %s", what, hipGetErrorString(e));
        return ELPH_E_HIP;
    }
    return ELPH_OK;
}

static unsigned xcd_grid(const elph_handle_s *h, int nvec) { return 8u * (unsigned)((h->L + 7) / 8) * (unsigned)nvec; }

int elph_fast_mul(elph_handle_s *h, int which, double *yS, const double *vS, int nvec) {
    ModelDev m = elph_model_dev(h);
    const size_t shm = 2 * (size_t)(h->npl * WAVE + 2 * WAVE) * sizeof(double);
    const dim3 grid(xcd_grid(h, nvec));
    const bool ssh = (h->kind == ELPH_MODEL_SSH);
    DISPATCH_NPL_F(h->npl, {
        if (ssh) {
            if (which == 0) hipLaunchKernelGGL((k_mul_fast<NPL, 0, true>), grid, dim3(WAVE), shm, h->stream, yS, vS, m);
            else if (which == 1) hipLaunchKernelGGL((k_mul_fast<NPL, 1, true>), grid, dim3(WAVE), shm, h->stream, yS, vS, m);
            else hipLaunchKernelGGL((k_mul_fast<NPL, 2, true>), grid, dim3(WAVE), shm, h->stream, yS, vS, m);
        } else {
            if (which == 0) hipLaunchKernelGGL((k_mul_fast<NPL, 0, false>), grid, dim3(WAVE), shm, h->stream, yS, vS, m);
            else if (which == 1) hipLaunchKernelGGL((k_mul_fast<NPL, 1, false>), grid, dim3(WAVE), shm, h->stream, yS, vS, m);
            else hipLaunchKernelGGL((k_mul_fast<NPL, 2, false>), grid, dim3(WAVE), shm, h->stream, yS, vS, m);
        }
    });
    return check_launch_f("k_mul_fast");
}

// slices per wave for the batched kernel: the largest T in {8,4,2} dividing L that still leaves >= 1024 waves
int elph_choose_T(const elph_handle_s *h, int nrhs) {
    if (!h->fast || h->force_T == 1) return 1;
    const int cand[3] = {8, 4, 2};
    for (int T : cand) {
        if (h->force_T > 1 && T != h->force_T) continue;
        if (h->L % T) continue;
        if (h->force_T > 1 || (int64_t)nrhs * (h->L / T) >= 1024) return T;
    }
    return 1;
}

int elph_fast_cg_ap(elph_handle_s *h, const CgBufs &B, int nrhs, int parity) {
    ModelDev m = elph_model_dev(h);
    const size_t shm = 2 * (size_t)(h->npl * WAVE + 2 * WAVE) * sizeof(double);
    const bool ssh = (h->kind == ELPH_MODEL_SSH);
    const int T = (B.npap == (int)h->L) ? 1 : (int)(h->L / B.npap);
    if (T > 1) {
        const dim3 grid((unsigned)(nrhs * (h->L / T)));
#define LAUNCH_CHUNK(TT)                                                                                           \
        DISPATCH_NPL_F(h->npl, {                                                                                       \
            if (ssh) hipLaunchKernelGGL((k_cg_ap_chunk<NPL, TT, true>), grid, dim3(WAVE), shm, h->stream, B, m, parity);  \
            else hipLaunchKernelGGL((k_cg_ap_chunk<NPL, TT, false>), grid, dim3(WAVE), shm, h->stream, B, m, parity);     \
        })
        if (T == 8) { LAUNCH_CHUNK(8); }
        else if (T == 4) { LAUNCH_CHUNK(4); }
        else { LAUNCH_CHUNK(2); }
#undef LAUNCH_CHUNK
        return check_launch_f("k_cg_ap_chunk");
    }
    const dim3 grid(xcd_grid(h, nrhs));
    DISPATCH_NPL_F(h->npl, {
        if (ssh) hipLaunchKernelGGL((k_cg_ap_fast<NPL, true>), grid, dim3(WAVE), shm, h->stream, B, m, parity);
        else hipLaunchKernelGGL((k_cg_ap_fast<NPL, false>), grid, dim3(WAVE), shm, h->stream, B, m, parity);
    });
    return check_launch_f("k_cg_ap_fast");
}

int elph_fast_cg_xr(elph_handle_s *h, const CgBufs &B, int nrhs, int parity) {
    const dim3 grid(xcd_grid(h, nrhs));
    DISPATCH_NPL_F(h->npl, {
        hipLaunchKernelGGL((k_cg_xr_fast<NPL>), grid, dim3(WAVE), 0, h->stream, B, (int)h->N, (int)h->L, parity);
    });
    return check_launch_f("k_cg_xr_fast");
}

int elph_fast_kpm_cheb(elph_handle_s *h, int nrhs, const CgState *st) {
    KpmDev K = elph_kpm_dev(h);
    ModelDev m = elph_model_dev(h);
    const int Lo2 = (int)((h->L + 1) / 2);
    static const bool complex_variant = []() { const char *e = getenv("ELPH_CHEB_COMPLEX"); return e && e[0] == '1'; }();
    if (complex_variant) {
        const size_t shm = (size_t)(h->npl * WAVE + 2 * WAVE) * sizeof(double2);
        DISPATCH_NPL_F(h->npl, {
            hipLaunchKernelGGL((k_kpm_cheb_fast<NPL>), dim3((unsigned)nrhs, (unsigned)Lo2), dim3(WAVE), shm, h->stream, h->d_nu, K,
                               m, Lo2, st);
        });
        return check_launch_f("k_kpm_cheb_fast");
    }
    static const bool no_sq = []() { const char *e = getenv("ELPH_NO_SQ"); return e && e[0] == '1'; }();
    if (h->sq_P > 0 && !no_sq) {
        if (h->sq_P == 2)
            hipLaunchKernelGGL((k_kpm_cheb_sq<2>), dim3((unsigned)nrhs, (unsigned)Lo2), dim3(2 * WAVE), 0, h->stream, h->d_nu, K,
                               h->d_sq_cbar, h->d_sq_sbar, (int)h->N, Lo2, st);
        else
            hipLaunchKernelGGL((k_kpm_cheb_sq<1>), dim3((unsigned)nrhs, (unsigned)Lo2), dim3(2 * WAVE), 0, h->stream, h->d_nu, K,
                               h->d_sq_cbar, h->d_sq_sbar, (int)h->N, Lo2, st);
        return check_launch_f("k_kpm_cheb_sq");
    }
    const size_t shm = (size_t)(2 * (h->npl * WAVE + 2 * WAVE) + 2 * h->npl * WAVE) * sizeof(double);
    DISPATCH_NPL_F(h->npl, {
        hipLaunchKernelGGL((k_kpm_cheb_ri<NPL>), dim3((unsigned)nrhs, (unsigned)Lo2), dim3(2 * WAVE), shm, h->stream, h->d_nu, K, m,
                           Lo2, st);
    });
    return check_launch_f("k_kpm_cheb_ri");
}

// ==========================================================================================
// Resident CG: the whole un-preconditioned solve in ONE launch, vectors held in registers.
// ==========================================================================================
// A wave owns T consecutive tau-slices of one right-hand side for the entire solve and keeps x, r, p (own slices
// plus one halo slice each side), z and exp(-dtau V) of those slices in registers; nothing of the Krylov vectors
// goes back to HBM between iterations.  Waves of one right-hand side meet twice per iteration through L2:
//   (1) p.z:  each wave publishes its partial, waits for all partials of its rhs, reduces them in the fixed order of
//       reduce_partials2  =>  alpha identical in every wave;
//   (2) r.r + halo:  each wave publishes the per-slice r.r partials of its slices AND its two boundary slices of the
//       new r, waits for all  =>  eps, kappa, stop test, beta identical in every wave; the neighbours' boundary
//       slices give p(t0-1), p(t0+T) of the next direction (p = r + beta p is pointwise, the old halo p is still in
//       registers) — the same "recompute the halo" trick as k_cg_ap_fast, without re-reading anything else.
// Everything the waves exchange (partials, boundary slices, flags) moves with device-scope (sc1) relaxed atomic
// loads/stores, which are coherent at the device level by themselves on gfx942/950; the order "data before flag" /
// "flag before data" is kept with s_waitcnt vmcnt(0) (a store is acknowledged once it is at the coherence point) —
// NOT with agent-scope release/acquire fences, whose L2 write-back + invalidate per meeting cost 0.4 us per wave
// (measured: 70 us per iteration at 160 waves).  Fast when the waves of a right-hand side share an XCD (their L2 is
// the meeting point): block b runs on XCD b % 8, so right-hand side r is given the blocks with b % 8 == r % 8.
// Arithmetic per element and the reduction trees are those of k_cg_ap_chunk<T> / k_cg_xr_fast: same iterates.
// Every spin is bounded; a wave that times out raises `abort`, every other wave sees it within 64 polls and leaves,
// and the host falls back to the two-kernel iteration (still on the GPU).
struct ResidentCtl {
    int *flagZ, *flagR;       // [nr][Wr] iteration counters (zeroed by the host before the launch)
    double *pz;               // [2][nr][Wr]
    double *rr;               // [2][nr][L]
    double *halo;             // [2][nr][Wr][2][NPL*64]
    int *abort;
    int rhs0, nr;
    long long spin_limit;
};

__device__ __forceinline__ int ld_flag(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ bool wait_all(const int *flags, int n, int target, int *abort, long long limit) {
    for (long long spin = 0;; ++spin) {
        int ok = 1;
        for (int i = threadIdx.x; i < n; i += WAVE) ok &= (ld_flag(flags + i) >= target);
        if (__all(ok)) break;
        if ((spin & 63) == 63 && ld_flag(abort) != 0) return false;
        if (spin > limit) {
            if (threadIdx.x == 0) __hip_atomic_store(abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
        __builtin_amdgcn_s_sleep(1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return true;
}

__device__ __forceinline__ void st_coh(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ void publish(int *flag, int value) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every sc1 store above has reached the coherence point
    if (threadIdx.x == 0) __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int NPL, int T>
__global__ void __launch_bounds__(WAVE) k_cg_resident(CgBufs B, ModelDev m, ResidentCtl R) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int NE = 4 * ((NPL + 1) / 2);
    constexpr int HS = NPL * WAVE;                    // halo slice stride
    double *bufA = lds, *bufB = lds + slab_len<NPL>();
    const int N = m.N, L = m.L, Wr = L / T;
    const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
    const int rq = k / Wr, w = k - rq * Wr;
    const int rl = rq * 8 + xcd;                      // right-hand side within this round
    if (rl >= R.nr) return;
    const int rhs = R.rhs0 + rl;
    const int t0 = w * T;
    const size_t ndim = (size_t)N * L;
    auto wrap = [L](int t) { return (t < 0) ? t + L : ((t >= L) ? t - L : t); };
    auto sgn = [](int t) { return (t == 0) ? -1.0 : 1.0; };
    const CgParams P = B.params;
    CgState *st2 = B.state + 2 * rhs;
    const CgState S = ld_state(st2);
    if (S.done || S.seq != 0) return;                 // only fresh solves (the host guarantees it)

    int sc[NPL];
#pragma unroll
    for (int q = 0; q < NPL; ++q) { const int s = threadIdx.x + q * WAVE; sc[q] = (s < N) ? s : N - 1; }
    double *xg = B.x + (size_t)rhs * ndim, *rg = B.r + (size_t)rhs * ndim;
    const double *pg = B.p + (size_t)rhs * ndim;      // parity 0: p0 of k_cg_init
    const double *Ech = m.E + (size_t)(rhs % m.nchains) * m.E_chain_stride;

    double x[T][NPL], r[T][NPL], z[T][NPL], p[T + 2][NPL], E[T + 1][NPL];
#pragma unroll
    for (int j = 0; j < T; ++j)
#pragma unroll
        for (int q = 0; q < NPL; ++q) {
            const size_t i = (size_t)(t0 + j) * N + sc[q];
            x[j][q] = xg[i]; r[j][q] = rg[i];
        }
#pragma unroll
    for (int j = 0; j < T + 2; ++j)
#pragma unroll
        for (int q = 0; q < NPL; ++q) p[j][q] = pg[(size_t)wrap(t0 + j - 1) * N + sc[q]];
#pragma unroll
    for (int j = 0; j <= T; ++j)
#pragma unroll
        for (int q = 0; q < NPL; ++q) E[j][q] = Ech[(size_t)wrap(t0 + j) * m.E_tau_stride + sc[q]];
    unsigned ij[NE];
    double cA[NE], sA[NE];
    lp_load_ij<NPL>(ij, m);
    lp_load_cs<NPL>(cA, sA, m.lp_c, m.lp_s);

    int *fZ = R.flagZ + (size_t)rl * Wr, *fR = R.flagR + (size_t)rl * Wr;
    const int wm = (w == 0) ? Wr - 1 : w - 1, wp = (w == Wr - 1) ? 0 : w + 1;
    double rho = S.rho, kmin = S.kmin, eps = S.eps;
    const double eps0 = S.eps0, normb = S.normb;

    for (long long seq = 0;; ++seq) {
        const int par = (int)(seq & 1);
        // ---- z = MtM p on the own slices (pipeline of k_cg_ap_chunk, operands in registers) -------------------
        double wprev[NPL], wcur[NPL];
#pragma unroll
        for (int q = 0; q < NPL; ++q) {
            const int s = threadIdx.x + q * WAVE;
            bufA[s] = E[0][q] * p[0][q];
            bufB[s] = E[1][q] * p[1][q];
        }
        WAVE_LDS_ORDER();
        lp_sweep<NPL, 2, false>(bufA, bufB, ij, cA, sA, cA, sA, m.ncol);
        {
            const double sga = sgn(t0), sgb = sgn(wrap(t0 + 1));
#pragma unroll
            for (int q = 0; q < NPL; ++q) {
                const int s = threadIdx.x + q * WAVE;
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
                const int s = threadIdx.x + q * WAVE;
                bufB[s] = wcur[q];
                if (more) bufA[s] = E[(j + 1 <= T) ? j + 1 : T][q] * p[j + 1][q];      // E(t0+j+1) .* p(t0+j)
            }
            WAVE_LDS_ORDER();
            lp_sweep_fr<NPL>(bufA, bufB, ij, cA, sA, cA, sA, m.ncol, more);
            const double sgj = sgn(wrap(t0 + j)), sgnn = sgn(wrap(t0 + j + 1));
            double wnext[NPL];
#pragma unroll
            for (int q = 0; q < NPL; ++q) {
                const int s = threadIdx.x + q * WAVE;
                const double zz = wprev[q] - sgj * E[j][q] * bufB[s];                   // z(t0+j-1)
                z[j - 1][q] = zz;
                if (s < N) acc += p[j][q] * zz;
                if (more) wnext[q] = p[(j + 2 <= T + 1) ? j + 2 : T + 1][q] - sgnn * bufA[s];   // w(t0+j+1)
            }
            WAVE_LDS_ORDER();
            if (more) {
#pragma unroll
                for (int q = 0; q < NPL; ++q) { wprev[q] = wcur[q]; wcur[q] = wnext[q]; }
            }
        }
        acc = wave_sum2(acc);
        // ---- meeting 1: p.z --------------------------------------------------------------------------------
        double *pzs = R.pz + ((size_t)par * R.nr + rl) * Wr;
        if (threadIdx.x == 0) st_coh(pzs + w, acc);
        publish(fZ + w, (int)seq + 1);
        if (!wait_all(fZ, Wr, (int)seq + 1, R.abort, R.spin_limit)) return;
        const double pap = reduce_partials2(pzs, Wr);
        const double alpha = rho / pap;
        // ---- x += alpha p, r -= alpha z, per-slice r.r; publish partials + boundary slices of r ----------------
        double *rrs = R.rr + ((size_t)par * R.nr + rl) * L;
        double *hal = R.halo + (((size_t)par * R.nr + rl) * Wr + w) * 2 * HS;
#pragma unroll
        for (int j = 0; j < T; ++j) {
            double a = 0.0;
#pragma unroll
            for (int q = 0; q < NPL; ++q) {
                const int s = threadIdx.x + q * WAVE;
                x[j][q] = x[j][q] + alpha * p[j + 1][q];
                const double rn = r[j][q] - alpha * z[j][q];
                r[j][q] = rn;
                if (s < N) a += rn * rn;
                if (j == 0) st_coh(hal + s, rn);
                if (T > 1 && j == T - 1) st_coh(hal + HS + s, rn);
            }
            a = wave_sum2(a);
            if (threadIdx.x == 0) st_coh(rrs + t0 + j, a);
        }
        publish(fR + w, (int)seq + 1);
        if (!wait_all(fR, Wr, (int)seq + 1, R.abort, R.spin_limit)) return;
        const double rr = reduce_partials2(rrs, L);
        // ---- stop test of iteration j = seq + 1 (IterativeSolvers.jl:286-295; same code as k_cg_ap_fast) -------
        const long long it = seq + 1;
        eps = sqrt(rr) / normb;
        const double qq = 2.0 * (double)it / log(2.0 * eps0 / eps);
        const double val = qq * qq;
        kmin = (val > kmin) ? val : kmin;
        int done = 0;
        if (eps < P.tol) done = 1;
        else if (kmin > P.kmax) done = 2;
        else if (it >= P.maxiter) done = 3;
        if (w == 0 && threadIdx.x == 0 && P.record_hist) B.hist[(size_t)rhs * P.hist_stride + it] = eps;
        if (done) {
#pragma unroll
            for (int j = 0; j < T; ++j)
#pragma unroll
                for (int q = 0; q < NPL; ++q) {
                    const int s = threadIdx.x + q * WAVE;
                    if (s < N) {
                        const size_t i = (size_t)(t0 + j) * N + s;
                        xg[i] = x[j][q]; rg[i] = r[j][q];
                    }
                }
            if (w == 0 && threadIdx.x == 0) {
                CgState o = S;
                o.rho = rho; o.kmin = kmin; o.eps = eps; o.seq = it + 1; o.iters = it; o.done = done;
                st2[0] = o;
                st2[1] = o;
            }
            return;
        }
        const double beta = rr / rho;
        rho = rr;
        // ---- next direction on own slices and on the two halo slices ----------------------------------------
        const double *hm = R.halo + (((size_t)par * R.nr + rl) * Wr + wm) * 2 * HS + ((T > 1) ? HS : 0);
        const double *hp = R.halo + (((size_t)par * R.nr + rl) * Wr + wp) * 2 * HS;
#pragma unroll
        for (int q = 0; q < NPL; ++q) {
            const int s = threadIdx.x + q * WAVE;
            p[0][q] = ld_coh(hm + s) + beta * p[0][q];
            p[T + 1][q] = ld_coh(hp + s) + beta * p[T + 1][q];
        }
#pragma unroll
        for (int j = 0; j < T; ++j)
#pragma unroll
            for (int q = 0; q < NPL; ++q) p[j + 1][q] = r[j][q] + beta * p[j + 1][q];
    }
}

// rhs per round for a given T (0: this T cannot run), from the occupancy of the kernel on this device
template <int NPL, int T>
static int resident_capacity(elph_handle_s *h, size_t shm) {
    int occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_cg_resident<NPL, T>, WAVE, shm) != hipSuccess) { (void)hipGetLastError(); return 0; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, h->device) != hipSuccess) { (void)hipGetLastError(); return 0; }
    const int cus_per_xcd = prop.multiProcessorCount / 8;
    if (cus_per_xcd < 1) return 0;
    const long long per_xcd = (long long)occ * cus_per_xcd * 3 / 4;      // head-room: the dispatcher need not pack perfectly
    const int Wr = (int)(h->L / T);
    return (int)(8 * (per_xcd / Wr));
}

template <int NPL>
static int res_cap_npl(elph_handle_s *h, size_t shm, int T) {
    if (T == 1) return resident_capacity<NPL, 1>(h, shm);
    if (T == 2) { if constexpr (NPL * 2 <= 8) return resident_capacity<NPL, 2>(h, shm); }
    if (T == 4) { if constexpr (NPL * 4 <= 8) return resident_capacity<NPL, 4>(h, shm); }
    return 0;
}

template <int NPL>
static void res_launch_npl(elph_handle_s *h, dim3 grid, size_t shm, int T, const CgBufs &B, const ModelDev &m, const ResidentCtl &R) {
    if (T == 1) hipLaunchKernelGGL((k_cg_resident<NPL, 1>), grid, dim3(WAVE), shm, h->stream, B, m, R);
    else if (T == 2) { if constexpr (NPL * 2 <= 8) hipLaunchKernelGGL((k_cg_resident<NPL, 2>), grid, dim3(WAVE), shm, h->stream, B, m, R); }
    else { if constexpr (NPL * 4 <= 8) hipLaunchKernelGGL((k_cg_resident<NPL, 4>), grid, dim3(WAVE), shm, h->stream, B, m, R); }
}

// Runs the whole un-preconditioned CG for rhs [0, nrhs) after elph_launch_cg_init.  *ran = false: not applicable
// (the caller uses the two-kernel iteration); ELPH_E_HIP with "resident" in the message: timed out (same fallback).
int elph_fast_cg_resident(elph_handle_s *h, const CgBufs &B, int nrhs, bool *ran) {
    *ran = false;
    // one-wave-per-slice-group variant: correct (bit-identical to the two-kernel path at equal T) but its 160-way
    // meetings cost more than two kernel boundaries (19 vs 9.6 us per iteration at config C) => opt-in only
    const char *eo = getenv("ELPH_RESIDENT_WAVES"), *et = getenv("ELPH_RESIDENT_T");     // read per solve: tests toggle them
    const bool off = !(eo && eo[0] == '1');
    const int forceT = et ? atoi(et) : 0;
    if (off || h->resident_broken || !h->fast || h->kind != ELPH_MODEL_HOLSTEIN || B.params.use_prec) return ELPH_OK;
    const int L = (int)h->L, npl = h->npl;
    const size_t shm = 2 * (size_t)(npl * WAVE + 2 * WAVE) * sizeof(double);
    int bestT = 0, bestCap = 0;
    const int cand[3] = {1, 2, 4};
    for (int T : cand) {
        if (L % T || npl * T > 8 || L / T < 2) continue;
        if (forceT && T != forceT) continue;
        int cap = 0;
        DISPATCH_NPL_F(npl, { cap = res_cap_npl<NPL>(h, shm, T); });
        if (cap <= 0) continue;
        if (bestT == 0 || (bestCap < nrhs && cap > bestCap)) { bestT = T; bestCap = cap; }
        if (bestCap >= nrhs) break;                   // the smallest T that takes the whole batch in one round
    }
    if (bestT == 0) return ELPH_OK;
    const int T = bestT, Wr = L / T, cap = bestCap;
    // control block (grown on demand)
    const int nr_max = std::min(nrhs, cap);
    const size_t HS = (size_t)npl * WAVE;
    const size_t n_flag = 2 * (size_t)nr_max * Wr, n_pz = 2 * (size_t)nr_max * Wr, n_rr = 2 * (size_t)nr_max * L,
                 n_halo = 2 * (size_t)nr_max * Wr * 2 * HS;
    const size_t need = (n_flag + 2) * sizeof(int) + (n_pz + n_rr + n_halo + 8) * sizeof(double);
    if (need > h->res_cap) {
        HIPCHK(hipStreamSynchronize(h->stream));
        if (h->d_res) HIPCHK(hipFree(h->d_res));
        h->d_res = nullptr;
        HIPCHK(hipMalloc(&h->d_res, need));
        h->res_cap = need;
    }
    ModelDev m = elph_model_dev(h);
    for (int rhs0 = 0; rhs0 < nrhs; rhs0 += cap) {
        const int nr = std::min(cap, nrhs - rhs0);
        ResidentCtl R;
        char *base = static_cast<char *>(h->d_res);
        R.pz = reinterpret_cast<double *>(base);
        R.rr = R.pz + 2 * (size_t)nr * Wr;
        R.halo = R.rr + 2 * (size_t)nr * L;
        R.flagZ = reinterpret_cast<int *>(R.halo + 2 * (size_t)nr * Wr * 2 * HS);
        R.flagR = R.flagZ + (size_t)nr * Wr;
        R.abort = R.flagR + (size_t)nr * Wr;
        R.rhs0 = rhs0; R.nr = nr;
        R.spin_limit = 1LL << 21;
        HIPCHK(hipMemsetAsync(R.flagZ, 0, (2 * (size_t)nr * Wr + 2) * sizeof(int), h->stream));
        const dim3 grid((unsigned)(8 * ((nr + 7) / 8) * Wr));
        DISPATCH_NPL_F(npl, { res_launch_npl<NPL>(h, grid, shm, T, B, m, R); });
        int rc = check_launch_f("k_cg_resident");
        if (rc) return rc;
        int ab = 0;
        HIPCHK(hipMemcpyAsync(&ab, R.abort, sizeof(int), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        if (ab) {
            h->resident_broken = true;
            elph_set_error("resident CG kernel timed out waiting for its peer waves (T=%d, %d rhs); falling back", T, nr);
            return ELPH_E_HIP;
        }
    }
    h->resident_T = T;
    *ran = true;
    return ELPH_OK;
}
