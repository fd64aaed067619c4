// kernels.hip — hand-written gfx950 (CDNA4, wave64) kernels of the fermion-force solver.
//
// These are the GENERIC kernels: any bond table (ragged colours, > 4 colours, arbitrary order) and any
// lattice up to 8192 sites.  One workgroup of BS = 64*W threads (W = ceil(N/512) wavefronts) owns one
// imaginary-time slice tau of one right-hand side.  Thread l holds sites l, l+BS, ... (NPL <= 8 per thread)
// in registers; the checkerboard sweep, which couples sites at fixed tau, goes through the workgroup's LDS
// slab with one barrier per colour (for W = 1 the compiler lowers it to a wave barrier).
// Lattices with N <= 512 and <= 4 colours take the latency-tuned single-wave kernels of cg_fast.hip instead.
// Grid = (L, nrhs) workgroups.
//
// Reference semantics: SURVEY.md Appendix A; file:line citations at each kernel.

#include <cstdlib>

#include "elph_internal.h"

#define WAVE ELPH_WAVE

// ------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------

__device__ __forceinline__ double wave_sum(double v) {
    // xor butterfly: every lane ends with the bit-identical total (a+b == b+a at every level),
    // and the tree is fixed => run-to-run deterministic.
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}

// every wave of the workgroup reduces ALL partials itself: same loads, same tree => same bits in every wave
__device__ __forceinline__ double reduce_partials(const double *p, int n) {
    double a = 0.0;
    for (int i = (threadIdx.x & (WAVE - 1)); i < n; i += WAVE) a += p[i];
    return wave_sum(a);
}

// sum over the whole workgroup (fixed order: wave tree, then waves 0..W-1); result valid in thread 0
__device__ __forceinline__ double block_sum(double v, double *scratch /* LDS, >= 16 doubles */) {
    v = wave_sum(v);
    const int nw = (blockDim.x + WAVE - 1) / WAVE;
    if (nw == 1) return v;
    __syncthreads();
    if ((threadIdx.x & (WAVE - 1)) == 0) scratch[threadIdx.x / WAVE] = v;
    __syncthreads();
    double t = 0.0;
    for (int w = 0; w < nw; ++w) t += scratch[w];
    return t;
}


// One checkerboard sweep over an LDS slab holding one tau-slice (Checkerboard.jl:57-83 forward,
// :149-175 transposed == colours in reverse order; bonds inside a colour are site-disjoint so
// their order is irrelevant).  c,s point at this slice's coefficients.
template <int NBUF, bool REVERSE>
__device__ __forceinline__ void cb_sweep(double *buf0, double *buf1, const double *c0, const double *s0,
                                         const double *c1, const double *s1, const ModelDev &m) {
    for (int cc = 0; cc < m.ncol; ++cc) {
        const int col = REVERSE ? (m.ncol - 1 - cc) : cc;
        const int b0 = m.coloff[col], b1 = m.coloff[col + 1];
        for (int n = b0 + threadIdx.x; n < b1; n += blockDim.x) {
            const int i = m.bi[n], j = m.bj[n];
            {
                const double cn = c0[n], sn = s0[n];
                const double t1 = buf0[i], t2 = buf0[j];
                buf0[i] = cn * t1 + sn * t2;
                buf0[j] = cn * t2 + sn * t1;
            }
            if (NBUF == 2) {
                const double cn = c1[n], sn = s1[n];
                const double t1 = buf1[i], t2 = buf1[j];
                buf1[i] = cn * t1 + sn * t2;
                buf1[j] = cn * t2 + sn * t1;
            }
        }
        __syncthreads();
    }
}

// the same with the bond program (i | j << 16, cosh, sinh per bond) resident in LDS: the recursion applies it order_w times, and
// reading four table entries per bond from global memory (L2 latency, every colour of every apply) was what the generic
// Chebyshev kernel spent its time on (2.5 - 5 us per apply at N = 576 ... 1024)
template <bool REVERSE>
__device__ __forceinline__ void cb_sweep_z_lds(double2 *buf, const unsigned *ij, const double *c, const double *s, const ModelDev &m) {
    for (int cc = 0; cc < m.ncol; ++cc) {
        const int col = REVERSE ? (m.ncol - 1 - cc) : cc;
        const int b0 = m.coloff[col], b1 = m.coloff[col + 1];
        for (int n = b0 + threadIdx.x; n < b1; n += blockDim.x) {
            const unsigned w = ij[n];
            const int i = (int)(w & 0xFFFFu), j = (int)(w >> 16);
            const double cn = c[n], sn = s[n];
            const double2 t1 = buf[i], t2 = buf[j];
            buf[i] = make_double2(cn * t1.x + sn * t2.x, cn * t1.y + sn * t2.y);
            buf[j] = make_double2(cn * t2.x + sn * t1.x, cn * t2.y + sn * t1.y);
        }
        __syncthreads();
    }
}

// complex variant for the KPM recursion (Checkerboard.jl:123-141,212-230 on complex N-vectors)
template <bool REVERSE>
__device__ __forceinline__ void cb_sweep_z(double2 *buf, const double *c, const double *s, const ModelDev &m) {
    for (int cc = 0; cc < m.ncol; ++cc) {
        const int col = REVERSE ? (m.ncol - 1 - cc) : cc;
        const int b0 = m.coloff[col], b1 = m.coloff[col + 1];
        for (int n = b0 + threadIdx.x; n < b1; n += blockDim.x) {
            const int i = m.bi[n], j = m.bj[n];
            const double cn = c[n], sn = s[n];
            const double2 t1 = buf[i], t2 = buf[j];
            buf[i] = make_double2(cn * t1.x + sn * t2.x, cn * t1.y + sn * t2.y);
            buf[j] = make_double2(cn * t2.x + sn * t1.x, cn * t2.y + sn * t1.y);
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// layout conversion  R: v[site*L + tau]  <->  S: v[tau*N + site]
// ------------------------------------------------------------------------------------------

// out[c*rows + r] = in[r*cols + c]; 32x32 tiles through LDS (+1 pad), blockDim (32,8), z = vector
__global__ void __launch_bounds__(256) k_transpose(double *__restrict__ out, const double *__restrict__ in,
                                                   int rows, int cols) {
    __shared__ double tile[32][33];
    const size_t base = (size_t)blockIdx.z * (size_t)rows * (size_t)cols;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (int k = threadIdx.y; k < 32; k += 8) {
        const int r = r0 + k, c = c0 + threadIdx.x;
        if (r < rows && c < cols) tile[k][threadIdx.x] = in[base + (size_t)r * cols + c];
    }
    __syncthreads();
    for (int k = threadIdx.y; k < 32; k += 8) {
        const int c = c0 + k, r = r0 + threadIdx.x;
        if (r < rows && c < cols) out[base + (size_t)c * rows + r] = tile[threadIdx.x][k];
    }
}

// update_model!(holstein), HolsteinModels.jl:526-549, fused with the R->S conversion:
// E_S[tau*N+i] = exp(-dtau*(lambda_i x + lambda2_i x^2 - mu_i)),  x in layout R.
__global__ void __launch_bounds__(256) k_expV(double *__restrict__ ES, const double *__restrict__ xR,
                                              const double *__restrict__ lam3, int N, int L, double dtau) {
    __shared__ double tile[32][33];
    const int t0 = blockIdx.x * 32, s0 = blockIdx.y * 32;   // in: rows = sites, cols = tau
    for (int k = threadIdx.y; k < 32; k += 8) {
        const int s = s0 + k, t = t0 + threadIdx.x;
        if (s < N && t < L) {
            const double x = xR[(size_t)s * L + t];
            tile[k][threadIdx.x] = exp(-dtau * (lam3[s] * x + lam3[N + s] * (x * x) + -lam3[2 * N + s]));
        }
    }
    __syncthreads();
    for (int k = threadIdx.y; k < 32; k += 8) {
        const int t = t0 + k, s = s0 + threadIdx.x;
        if (s < N && t < L) ES[(size_t)t * N + s] = tile[threadIdx.x][k];
    }
}

__global__ void k_zero(double *p, long long n) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0.0;
}

// ------------------------------------------------------------------------------------------
// M v, M^T v, M^T M v   (HolsteinModels.jl:569-684, SSHModels.jl:581-701, Models.jl:215-224)
//   (M v)(t)   = v(t) - sg(t)   CB(t) [E(t) .* v(t-1)],        sg(0) = -1 else +1
//   (M^T v)(t) = v(t) - sg(t+1) E(t+1) .* [CB(t+1)^T v(t+1)]
//   M^T M in ONE pass: a wave owning slice t rebuilds w(t) and w(t+1) (w = M v) from
//   v(t-1..t+1) and applies the transposed sweep to w(t+1) — no v' round trip through memory.
// ------------------------------------------------------------------------------------------

template <int NPL, int WHICH>
__global__ void __launch_bounds__(1024) k_mul(double *__restrict__ y, const double *__restrict__ v, ModelDev m) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *bufA = lds, *bufB = lds + m.N;
    const int N = m.N, L = m.L;
    const int t = blockIdx.x;
    const int tm1 = (t == 0) ? L - 1 : t - 1;
    const int tp1 = (t == L - 1) ? 0 : t + 1;
    const size_t vec = (size_t)blockIdx.y * (size_t)N * (size_t)L;
    const double *vv = v + vec;
    double *yy = y + vec;
    const double sg0 = (t == 0) ? -1.0 : 1.0, sg1 = (tp1 == 0) ? -1.0 : 1.0;
    ssh_chain_select(m, (int)blockIdx.y);
    const double *c0 = m.c + (size_t)t * m.cs_tau_stride, *s0 = m.s + (size_t)t * m.cs_tau_stride;
    const double *c1 = m.c + (size_t)tp1 * m.cs_tau_stride, *s1 = m.s + (size_t)tp1 * m.cs_tau_stride;
    const double *Ech = m.E + (size_t)(blockIdx.y % m.nchains) * m.E_chain_stride;
    const double *E0 = Ech + (size_t)t * m.E_tau_stride, *E1 = Ech + (size_t)tp1 * m.E_tau_stride;

    if (WHICH == 0) {  // y = M v
        double v0[NPL];
#pragma unroll
        for (int q = 0; q < NPL; ++q) {
            const int s = threadIdx.x + q * blockDim.x;
            if (s < N) {
                v0[q] = vv[(size_t)t * N + s];
                bufA[s] = E0[s] * vv[(size_t)tm1 * N + s];
            }
        }
        __syncthreads();
        cb_sweep<1, false>(bufA, nullptr, c0, s0, nullptr, nullptr, m);
#pragma unroll
        for (int q = 0; q < NPL; ++q) {
            const int s = threadIdx.x + q * blockDim.x;
            if (s < N) yy[(size_t)t * N + s] = v0[q] - sg0 * bufA[s];
        }
    } else if (WHICH == 1) {  // y = M^T v
        double v0[NPL];
#pragma unroll
        for (int q = 0; q < NPL; ++q) {
            const int s = threadIdx.x + q * blockDim.x;
            if (s < N) {
                v0[q] = vv[(size_t)t * N + s];
                bufA[s] = vv[(size_t)tp1 * N + s];
            }
        }
        __syncthreads();
        cb_sweep<1, true>(bufA, nullptr, c1, s1, nullptr, nullptr, m);
#pragma unroll
        for (int q = 0; q < NPL; ++q) {
            const int s = threadIdx.x + q * blockDim.x;
            if (s < N) yy[(size_t)t * N + s] = v0[q] - sg1 * E1[s] * bufA[s];
        }
    } else {  // y = M^T M v
        double v0[NPL], vp[NPL], w0[NPL], e1[NPL];
#pragma unroll
        for (int q = 0; q < NPL; ++q) {
            const int s = threadIdx.x + q * blockDim.x;
            if (s < N) {
                const double vm = vv[(size_t)tm1 * N + s];
                v0[q] = vv[(size_t)t * N + s];
                vp[q] = vv[(size_t)tp1 * N + s];
                e1[q] = E1[s];
                bufA[s] = E0[s] * vm;
                bufB[s] = e1[q] * v0[q];
            }
        }
        __syncthreads();
        cb_sweep<2, false>(bufA, bufB, c0, s0, c1, s1, m);
#pragma unroll
        for (int q = 0; q < NPL; ++q) {
            const int s = threadIdx.x + q * blockDim.x;
            if (s < N) {
                w0[q] = v0[q] - sg0 * bufA[s];
                bufB[s] = vp[q] - sg1 * bufB[s];
            }
        }
        __syncthreads();
        cb_sweep<1, true>(bufB, nullptr, c1, s1, nullptr, nullptr, m);
#pragma unroll
        for (int q = 0; q < NPL; ++q) {
            const int s = threadIdx.x + q * blockDim.x;
            if (s < N) yy[(size_t)t * N + s] = w0[q] - sg1 * e1[q] * bufB[s];
        }
    }
}

// ------------------------------------------------------------------------------------------
// Conjugate gradient (IterativeSolvers.jl:153-234 with P, :239-314 without)
// Two kernels per iteration, no host involvement, all control flow from device-resident state:
//   k_cg_ap : [stop test of the previous iteration] beta; p = (z|r) + beta p; z = MtM p; partial p.z
//   k_cg_xr : alpha; x += alpha p; r -= alpha z; partial r.r
// Global reductions = per-slice partials written by one kernel and re-reduced, in a fixed order,
// by every wave of the next kernel (deterministic; no atomics; no extra launch).
// ------------------------------------------------------------------------------------------


// PX (round 6: the p/x-fused preconditioned iteration for every lattice of the generic family): the search direction arrives READY in B.p slot 0
// (the inverse tau-transform formed p = P^-1 r + beta p and applied x += alpha p in its epilogue, dft_mfma.hip: PxFuse) — no P^-1 r, no p_new.
template <int NPL, bool PX = false>
__global__ void __launch_bounds__(1024) k_cg_ap(CgBufs B, ModelDev m, int parity) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *bufA = lds, *bufB = lds + m.N;
    const int N = m.N, L = m.L;
    const int t = blockIdx.x, rhs = blockIdx.y;
    const size_t ndim = (size_t)N * L;
    // state copy of this launch = st2[parity]; the other copy is written at the end (no same-kernel reader)
    CgState *st2 = B.state + 2 * rhs;
    const CgState S = st2[parity];
    CgState *Sout = st2 + (parity ^ 1);
    if (S.done) {
        if (t == 0 && threadIdx.x == 0) *Sout = S;
        return;
    }
    const CgParams P = B.params;
    const long long seq = S.seq;          // == completed iterations so far
    const bool first = (seq == 0);

    double beta = 0.0, rho = S.rho, kmin = S.kmin, eps = S.eps;
    if (!first) {
        // stop test of iteration `seq` (IterativeSolvers.jl:286-295 / :211-219)
        const double rr = reduce_partials(B.rr + (size_t)rhs * L, L);
        eps = sqrt(rr) / S.normb;
        const double q = 2.0 * (double)seq / log(2.0 * S.eps0 / eps);
        const double val = q * q;
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
        const double rho_new = P.use_prec ? reduce_partials(B.rz + (size_t)rhs * B.nrz, B.nrz) : rr;
        beta = rho_new / S.rho;            // :222-223 / :303-304
        rho = rho_new;
    }

    const double *src = (P.use_prec ? B.zp : B.r) + (size_t)rhs * ndim;
    const double *pold = B.p + ((size_t)(PX ? 0 : parity) * B.nrhs + rhs) * ndim;
    double *pnew = B.p + ((size_t)(parity ^ 1) * B.nrhs + rhs) * ndim;
    double *z = B.z + (size_t)rhs * ndim;

    const int tm1 = (t == 0) ? L - 1 : t - 1;
    const int tp1 = (t == L - 1) ? 0 : t + 1;
    const double sg0 = (t == 0) ? -1.0 : 1.0, sg1 = (tp1 == 0) ? -1.0 : 1.0;
    ssh_chain_select(m, rhs);
    const double *c0 = m.c + (size_t)t * m.cs_tau_stride, *s0 = m.s + (size_t)t * m.cs_tau_stride;
    const double *c1 = m.c + (size_t)tp1 * m.cs_tau_stride, *s1 = m.s + (size_t)tp1 * m.cs_tau_stride;
    const double *Ech = m.E + (size_t)(rhs % m.nchains) * m.E_chain_stride;
    const double *E0 = Ech + (size_t)t * m.E_tau_stride, *E1 = Ech + (size_t)tp1 * m.E_tau_stride;

    double p0[NPL], pp[NPL], w0[NPL], e1[NPL];
#pragma unroll
    for (int q = 0; q < NPL; ++q) {
        const int s = threadIdx.x + q * blockDim.x;
        if (s < N) {
            const size_t im = (size_t)tm1 * N + s, i0 = (size_t)t * N + s, ip = (size_t)tp1 * N + s;
            double pm;
            if (PX || first) {                 // p0 = z0|r0 was stored by the init kernel (PX: the ready p of every iteration)
                pm = pold[im]; p0[q] = pold[i0]; pp[q] = pold[ip];
            } else {                           // p = (z|r) + beta p   (:229-230 / :309-310)
                pm = src[im] + beta * pold[im];
                p0[q] = src[i0] + beta * pold[i0];
                pp[q] = src[ip] + beta * pold[ip];
            }
            if (!PX) pnew[i0] = p0[q];
            e1[q] = E1[s];
            bufA[s] = E0[s] * pm;
            bufB[s] = e1[q] * p0[q];
        }
    }
    __syncthreads();
    cb_sweep<2, false>(bufA, bufB, c0, s0, c1, s1, m);
#pragma unroll
    for (int q = 0; q < NPL; ++q) {
        const int s = threadIdx.x + q * blockDim.x;
        if (s < N) {
            w0[q] = p0[q] - sg0 * bufA[s];
            bufB[s] = pp[q] - sg1 * bufB[s];
        }
    }
    __syncthreads();
    cb_sweep<1, true>(bufB, nullptr, c1, s1, nullptr, nullptr, m);
    double acc = 0.0;
#pragma unroll
    for (int q = 0; q < NPL; ++q) {
        const int s = threadIdx.x + q * blockDim.x;
        if (s < N) {
            const double zz = w0[q] - sg1 * e1[q] * bufB[s];
            z[(size_t)t * N + s] = zz;
            if (s >= B.dot_lo && s < B.dot_hi) acc += p0[q] * zz;
        }
    }
    acc = block_sum(acc, lds + 2 * (size_t)N);
    if (threadIdx.x == 0) {
        B.pap[(size_t)rhs * B.npap + t] = acc;
        if (t == 0) {
            CgState o = S;
            o.rho = rho; o.kmin = kmin; o.eps = eps; o.seq = seq + 1; o.iters = seq; o.done = 0;
            *Sout = o;
        }
    }
}

template <int NPL>
__global__ void __launch_bounds__(1024) k_cg_xr(CgBufs B, int N, int L, int parity) {
    const int t = blockIdx.x, rhs = blockIdx.y;
    const size_t ndim = (size_t)N * L;
    const CgState S = B.state[2 * rhs + parity];
    if (S.done) return;
    const double pap = reduce_partials(B.pap + (size_t)rhs * B.npap, B.npap);
    const double alpha = S.rho / pap;                       // :202 / :279
    const double *p = B.p + ((size_t)parity * B.nrhs + rhs) * ndim;
    const double *z = B.z + (size_t)rhs * ndim;
    double *x = B.x + (size_t)rhs * ndim, *r = B.r + (size_t)rhs * ndim;
    double acc = 0.0;
#pragma unroll
    for (int q = 0; q < NPL; ++q) {
        const int s = threadIdx.x + q * blockDim.x;
        if (s < N) {
            const size_t i = (size_t)t * N + s;
            x[i] += alpha * p[i];                           // :205 / :282
            const double rn = r[i] - alpha * z[i];          // :208 / :285
            r[i] = rn;
            if (s >= B.dot_lo && s < B.dot_hi) acc += rn * rn;
        }
    }
    __shared__ double scratch[16];
    acc = block_sum(acc, scratch);
    if (threadIdx.x == 0) B.rr[(size_t)rhs * L + t] = acc;
}

// r0 = b - A x0 (A x0 precomputed into tmp), p0 = r0 (un-preconditioned), partial r.r and b.b
__global__ void __launch_bounds__(WAVE) k_cg_init(CgBufs B, const double *__restrict__ b,
                                                  const double *__restrict__ Ax, double *bb, int N, int L) {
    const int t = blockIdx.x, rhs = blockIdx.y;
    const size_t ndim = (size_t)N * L;
    const double *bv = b + (size_t)rhs * ndim, *ax = Ax + (size_t)rhs * ndim;
    double *r = B.r + (size_t)rhs * ndim;
    double *p0 = B.p + (size_t)rhs * ndim;    // seq 0 reads p[0]
    double a = 0.0, c = 0.0;
    for (int s = threadIdx.x; s < N; s += WAVE) {
        const size_t i = (size_t)t * N + s;
        const double bi = bv[i];
        const double ri = 1.0 * bi + -1.0 * ax[i];          // axpby!(1,b,-1,r), :179-180 / :262-263
        r[i] = ri;
        p0[i] = ri;
        if (s >= B.dot_lo && s < B.dot_hi) {
            a += ri * ri;
            c += bi * bi;
        }
    }
    a = wave_sum(a);
    c = wave_sum(c);
    if (threadIdx.x == 0) {
        B.rr[(size_t)rhs * L + t] = a;
        bb[(size_t)rhs * L + t] = c;
    }
}

// preconditioned start: p0 = z0 = P^-1 r0 (zp already computed, r.z partials in B.rz)
__global__ void __launch_bounds__(WAVE) k_cg_init_prec(CgBufs B, int N, int L) {
    const int t = blockIdx.x, rhs = blockIdx.y;
    const size_t ndim = (size_t)N * L;
    const double *zp = B.zp + (size_t)rhs * ndim;
    double *p0 = B.p + (size_t)rhs * ndim;
    for (int s = threadIdx.x; s < N; s += WAVE) {
        const size_t i = (size_t)t * N + s;
        p0[i] = zp[i];
    }
}

// one wave per rhs: seed the state (normb, eps0, rho0) — IterativeSolvers.jl:176-195 / :259-274
__global__ void __launch_bounds__(WAVE) k_cg_state0(CgBufs B, const double *bb, int L) {
    const int rhs = blockIdx.x;
    const CgParams P = B.params;
    const double rr = reduce_partials(B.rr + (size_t)rhs * L, L);
    const double nb2 = reduce_partials(bb + (size_t)rhs * L, L);
    const double rho = P.use_prec ? reduce_partials(B.rz + (size_t)rhs * B.nrz, B.nrz) : rr;
    if (threadIdx.x == 0) {
        CgState o;
        o.normb = sqrt(nb2);
        o.eps0 = sqrt(rr) / o.normb;
        o.eps = o.eps0;
        o.rho = rho;
        o.kmin = 0.0;
        o.seq = 0; o.iters = 0; o.done = 0; o.pad = 0;
        B.state[2 * rhs] = o;
        o.seq = -1;
        B.state[2 * rhs + 1] = o;
        if (P.record_hist) B.hist[(size_t)rhs * P.hist_stride] = o.eps0;
    }
}

// true residual |A x - b| / |b| (Models.jl:94-97,150-154): partials of |Ax-b|^2 and |b|^2
__global__ void __launch_bounds__(WAVE) k_resid_part(const double *__restrict__ Ax, const double *__restrict__ b,
                                                     double *pa, double *pb, int N, int L) {
    const int t = blockIdx.x, rhs = blockIdx.y;
    const size_t ndim = (size_t)N * L;
    double a = 0.0, c = 0.0;
    for (int s = threadIdx.x; s < N; s += WAVE) {
        const size_t i = (size_t)rhs * ndim + (size_t)t * N + s;
        const double d = Ax[i] - b[i];
        a += d * d;
        c += b[i] * b[i];
    }
    a = wave_sum(a);
    c = wave_sum(c);
    if (threadIdx.x == 0) {
        pa[(size_t)rhs * L + t] = a;
        pb[(size_t)rhs * L + t] = c;
    }
}

__global__ void __launch_bounds__(WAVE) k_resid_final(const double *pa, const double *pb, double *out, int L) {
    const int rhs = blockIdx.x;
    const double a = reduce_partials(pa + (size_t)rhs * L, L);
    const double c = reduce_partials(pb + (size_t)rhs * L, L);
    if (threadIdx.x == 0) out[rhs] = sqrt(a) / sqrt(c);
}

// ------------------------------------------------------------------------------------------
// tau-axis transforms: see dft.hip.  Only the two API helpers for the complex tau_to_omega!/omega_to_tau!
// entry points live here (tw2[m] = exp(-i pi m / L), m in [0, 2L)).
// ------------------------------------------------------------------------------------------

// full complex half-spectrum -> full spectrum expansion for the tau_to_omega API (complex output, layout S)
__global__ void k_expand_spectrum(double2 *__restrict__ full, const double2 *__restrict__ half, int N, int L, int Lo2) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)N * L) return;
    const int k = (int)(i / N), s = (int)(i % N);
    double2 vv;
    if (k < Lo2) vv = half[(size_t)k * N + s];
    else {
        vv = half[(size_t)(L - 1 - k) * N + s];
        vv.y = -vv.y;
    }
    full[i] = vv;
}

// generic complex inverse twisted transform for omega_to_tau on arbitrary (non-symmetric) spectra:
// out[t][s] = Re( conj(Theta_t) (1/L) sum_k exp(2 pi i k t / L) nu[k][s] )
__global__ void __launch_bounds__(WAVE) k_dft_inv_twisted_full(double *__restrict__ out, const double2 *__restrict__ nu,
                                                               const double2 *__restrict__ tw2, int N, int L) {
    const int s = blockIdx.x * WAVE + threadIdx.x;
    const int t = blockIdx.y;
    const int twoL = 2 * L;
    double acc = 0.0;
    int m = t % twoL;
    for (int k = 0; k < L; ++k) {
        const double2 x = (s < N) ? nu[(size_t)k * N + s] : make_double2(0.0, 0.0);
        const double2 w = tw2[m];
        acc += w.x * x.x + w.y * x.y;
        m += 2 * t;
        if (m >= twoL) m -= twoL;
        if (m >= twoL) m -= twoL;
    }
    if (s < N) out[(size_t)t * N + s] = acc / (double)L;
}

// ------------------------------------------------------------------------------------------
// KPM preconditioner, per-omega Chebyshev recursion (KPMPreconditioners.jl:606-693,758-778)
// One wave per frequency block; u_{n-1}, u_n and the accumulator live in registers, the
// checkerboard exchange goes through a complex LDS slab.  Blocks are scheduled longest-order first.
// ------------------------------------------------------------------------------------------

template <int NPL, bool TRANSPOSED>
__device__ __forceinline__ void kpm_mulAprime(double2 (&out)[NPL], const double2 (&un)[NPL], double2 *buf,
                                              const double (&eb)[NPL], double a, double b, const KpmDev &K,
                                              const ModelDev &m, const unsigned *K_lds_ij, const double *K_lds_c,
                                              const double *K_lds_s) {
    const int N = m.N;
#pragma unroll
    for (int q = 0; q < NPL; ++q) {
        const int s = threadIdx.x + q * blockDim.x;
        if (s < N) buf[s] = TRANSPOSED ? un[q] : make_double2(eb[q] * un[q].x, eb[q] * un[q].y);
    }
    __syncthreads();
    if (K_lds_ij) cb_sweep_z_lds<TRANSPOSED>(buf, K_lds_ij, K_lds_c, K_lds_s, m);
    else cb_sweep_z<TRANSPOSED>(buf, K.cbar, K.sbar, m);
#pragma unroll
    for (int q = 0; q < NPL; ++q) {
        const int s = threadIdx.x + q * blockDim.x;
        if (s < N) {
            double2 av = buf[s];
            if (TRANSPOSED) { av.x *= eb[q]; av.y *= eb[q]; }
            out[q] = make_double2(a * av.x - b * un[q].x, a * av.y - b * un[q].y);   // :685-693
        }
    }
    __syncthreads();
}

template <int NPL, bool TRANSPOSED, bool CONJ>
__device__ __forceinline__ void kpm_series(double2 (&acc)[NPL], const double2 (&vin)[NPL], double2 *buf,
                                           const double (&eb)[NPL], const double2 *c, int order, const KpmDev &K,
                                           const ModelDev &m, double a, double b, const unsigned *lij, const double *lc,
                                           const double *ls) {
    double2 um1[NPL], un[NPL], up1[NPL];
    {
        double2 c0 = c[0];
        if (CONJ) c0.y = -c0.y;
#pragma unroll
        for (int q = 0; q < NPL; ++q) {
            acc[q] = make_double2(c0.x * vin[q].x - c0.y * vin[q].y, c0.x * vin[q].y + c0.y * vin[q].x);
            un[q] = vin[q];
            um1[q] = make_double2(0.0, 0.0);
        }
    }
    if (order > 1) {
        kpm_mulAprime<NPL, TRANSPOSED>(up1, un, buf, eb, a, b, K, m, lij, lc, ls);
        for (int n = 2;; ++n) {
#pragma unroll
            for (int q = 0; q < NPL; ++q) { um1[q] = un[q]; un[q] = up1[q]; }
            double2 cn = c[n - 1];
            if (CONJ) cn.y = -cn.y;
#pragma unroll
            for (int q = 0; q < NPL; ++q) {
                acc[q].x += cn.x * un[q].x - cn.y * un[q].y;
                acc[q].y += cn.x * un[q].y + cn.y * un[q].x;
            }
            if (n == order) break;
            kpm_mulAprime<NPL, TRANSPOSED>(up1, un, buf, eb, a, b, K, m, lij, lc, ls);
#pragma unroll
            for (int q = 0; q < NPL; ++q) {
                up1[q].x = 2.0 * up1[q].x - um1[q].x;
                up1[q].y = 2.0 * up1[q].y - um1[q].y;
            }
        }
    }
}

template <int NPL>
// rz_part != nullptr (round 6, the p/x-fused iteration): the block's share of r.(P^-1 r) in frequency space (Parseval for the twisted transform:
// a.b = (1/L) sum_k conj(a_k) b_k, the mirror frequency L-1-k contributing the same) into slot blockIdx.y of this right-hand side, the slots
// beyond Lo2 cleared by the blocks in turn; a chain whose expansion is inactive hands over the r.r partials of the residual update instead.
__global__ void __launch_bounds__(1024) k_kpm_cheb(double2 *__restrict__ nu, KpmDev K, ModelDev m, int Lo2,
                                                   const CgState *state, int lds_tables, double *__restrict__ rz_part, int nrz,
                                                   const double *__restrict__ rr_part) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double2 *buf = reinterpret_cast<double2 *>(lds);
    const int rhs = blockIdx.x;   // x = right-hand side, y = frequency in longest-first order: ALL long recursions are dispatched first
    if (state && state[2 * rhs].done) return;   // `state` points at the current copy
    // bond program -> LDS (after the slab: [N] double2 | [nb] cosh | [nb] sinh | [nb] i | j << 16)
    const unsigned *lij = nullptr;
    const double *lc = nullptr, *ls = nullptr;
    const KpmChainView V = kpm_chain_view(K, rhs, m.N);
    K.cbar = V.cbar; K.sbar = V.sbar;                               // this chain's averaged hopping (SSH chains) — BEFORE the tables are staged:
    if (lds_tables) {                                               // (round 5: the LDS copy took chain 0's tables for every chain of an SSH batch)
        double *tc = lds + 2 * (size_t)m.N, *ts = tc + m.nb;
        unsigned *tij = reinterpret_cast<unsigned *>(ts + m.nb);
        for (int n = threadIdx.x; n < m.nb; n += blockDim.x) {
            tc[n] = K.cbar[n]; ts[n] = K.sbar[n];
            tij[n] = (unsigned)m.bi[n] | ((unsigned)m.bj[n] << 16);
        }
        lij = tij; lc = tc; ls = ts;                               // the first barrier inside kpm_mulAprime publishes them
    }
    const int w = V.wsched[blockIdx.y];
    const int N = m.N;
    const int order = V.order[w];
    const double2 *c = K.coeff + V.coff[w];
    double2 *u = nu + ((size_t)rhs * Lo2 + w) * N;
    double2 vin[NPL], mid[NPL], res[NPL];
    double eb[NPL];
#pragma unroll
    for (int q = 0; q < NPL; ++q) {
        const int s = threadIdx.x + q * blockDim.x;
        vin[q] = (s < N) ? u[s] : make_double2(0.0, 0.0);
        eb[q] = (s < N) ? V.Ebar[s] : 0.0;
    }
    kpm_series<NPL, true, true>(mid, vin, buf, eb, c, order, K, m, V.a, V.b, lij, lc, ls);     // M^-T[w,w], conj coefficients (:621-648)
    kpm_series<NPL, false, false>(res, mid, buf, eb, c, order, K, m, V.a, V.b, lij, lc, ls);   // M^-1[w,w]                     (:650-677)
    double dot = 0.0;
#pragma unroll
    for (int q = 0; q < NPL; ++q) {
        const int s = threadIdx.x + q * blockDim.x;
        if (s < N) { u[s] = res[q]; dot += vin[q].x * res[q].x + vin[q].y * res[q].y; }
    }
    if (rz_part) {
        dot = block_sum(dot, lds);           // (the slab is free: both series are done)
        if (threadIdx.x == 0) {
            const int Ltau = m.L, bid = (int)blockIdx.y;
            double *slots = rz_part + (size_t)rhs * nrz;
            const double wgt = ((Ltau & 1) && w == Lo2 - 1) ? 1.0 : 2.0;
            if (V.active) {
                slots[bid] = wgt * dot / (double)Ltau;
                for (int qq = Lo2 + bid; qq < nrz; qq += Lo2) slots[qq] = 0.0;
            } else {
                for (int qq = bid; qq < nrz; qq += Lo2) slots[qq] = (qq < Ltau) ? rr_part[(size_t)rhs * Ltau + qq] : 0.0;
            }
        }
    }
}

// Ebar[i] = mean_tau E[tau][i]  (update_A!, KPMPreconditioners.jl:332-349)
// update_model! of the SSH model on the device (SSHModels.jl:510-562).
//   pass 1 (fill):   every (tau, bond) gets cosh/sinh of its bare hopping  (bonds without a phonon keep that)
//   pass 2 (fields): phonon field (p, tau):  t' = t - (alpha x + sign(x) alpha2 x^2),  cosh/sinh(dtau t') at the bond's
//                    checkerboard position — in the tau-major table AND in the lane-program copy of the fast kernels
// par = [t | alpha | alpha2] per phonon; cb0 = 0-based checkerboard position of each phonon's bond; slot = lane-program
// slot of each checkerboard bond (-1: generic kernels only); x in the reference's field order (tau fastest).
__global__ void __launch_bounds__(256) k_ssh_fill(double *__restrict__ c, double *__restrict__ s, double *__restrict__ lpc,
                                                  double *__restrict__ lps, const double *__restrict__ tbare,
                                                  const int *__restrict__ slot, int nb, int L, int lp_stride, double dtau) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)nb * L) return;
    { const size_t ch = blockIdx.y; c += ch * (size_t)nb * L; s += ch * (size_t)nb * L; lpc += ch * (size_t)L * lp_stride; lps += ch * (size_t)L * lp_stride; }
    const int n = (int)(i % nb), t = (int)(i / nb);
    const double a = dtau * tbare[n], cc = cosh(a), ss = sinh(a);
    c[i] = cc; s[i] = ss;
    if (slot && slot[n] >= 0) { lpc[(size_t)t * lp_stride + slot[n]] = cc; lps[(size_t)t * lp_stride + slot[n]] = ss; }
}

__global__ void __launch_bounds__(256) k_ssh_fields(double *__restrict__ c, double *__restrict__ s, double *__restrict__ lpc,
                                                    double *__restrict__ lps, const double *__restrict__ x,
                                                    const double *__restrict__ par, const int *__restrict__ cb0,
                                                    const int *__restrict__ slot, int nph, int nb, int L, int lp_stride,
                                                    double dtau, int x_tau_major) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)nph * L) return;
    { const size_t ch = blockIdx.y; c += ch * (size_t)nb * L; s += ch * (size_t)nb * L; lpc += ch * (size_t)L * lp_stride; lps += ch * (size_t)L * lp_stride;
      x += ch * (size_t)nph * L; }
    const int t = (int)(i % L), p = (int)(i / L);           // field index = (phonon - 1) Ltau + tau  (Utilities.jl:12-15)
    const double xt = x_tau_major ? x[(size_t)t * nph + p] : x[i];   // tau-major: the HMC trajectory's device layout
    const double sg = (xt > 0.0) ? 1.0 : ((xt < 0.0) ? -1.0 : 0.0);
    const double v = par[nph + p] * xt + sg * par[2 * nph + p] * (xt * xt);
    const double a = dtau * (par[p] - v), cc = cosh(a), ss = sinh(a);
    const int n = cb0[p];
    c[(size_t)t * nb + n] = cc; s[(size_t)t * nb + n] = ss;
    if (slot && slot[n] >= 0) { lpc[(size_t)t * lp_stride + slot[n]] = cc; lps[(size_t)t * lp_stride + slot[n]] = ss; }
}

// blockIdx.y = chain: every chain has its own exp(dtau mu) (the same values unless the chemical potential is tuned per chain)
__global__ void __launch_bounds__(256) k_ssh_expmu(double *__restrict__ E, const double *__restrict__ mu, int N, double dtau,
                                                   int mu_stride) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < N) E[(size_t)blockIdx.y * N + i] = exp(dtau * mu[(size_t)blockIdx.y * mu_stride + i]);
}

// F[(p, tau)] = sg(tau) dtau (alpha_p + 2 alpha2_p x) q[tau][bond(p)]  — dMdx of the bond-phonon fields (SSHModels.jl:797-823)
__global__ void __launch_bounds__(256) k_ssh_scatter(double *__restrict__ F, const double *__restrict__ q, const double *__restrict__ x,
                                                     const double *__restrict__ par, const int *__restrict__ cb0, int nph, int nb,
                                                     int L, double dtau, int tau_major, double scale) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)nph * L) return;
    { const size_t ch = blockIdx.y; F += ch * (size_t)nph * L; x += ch * (size_t)nph * L; q += ch * (size_t)L * nb; }   // chain
    const int t = (int)(i % L), p = (int)(i / L);
    const size_t k = tau_major ? (size_t)t * nph + p : (size_t)i;             // x and F share one layout
    const double sg = (t == 0) ? -1.0 : 1.0;                                  // "flip sign if τ=1" (:809-811)
    const double dKdx = par[nph + p] + 2.0 * par[2 * nph + p] * x[k];         // ∂K/∂x as the reference takes it (:803)
    F[k] = scale * sg * dtau * dKdx * q[(size_t)t * nb + cb0[p]];
}

// tau-means of the SSH cosh/sinh tables (update_A!, KPMPreconditioners.jl:355-381): one thread per bond
__global__ void __launch_bounds__(256) k_cs_bar(double *__restrict__ cbar, double *__restrict__ sbar, const double *__restrict__ c,
                                                const double *__restrict__ s, int nb, int L) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= nb) return;
    { const size_t ch = blockIdx.y; c += ch * (size_t)L * nb; s += ch * (size_t)L * nb; cbar += ch * (size_t)nb; sbar += ch * (size_t)nb; }
    double a = 0.0, b = 0.0;
    for (int t = 0; t < L; ++t) { a += c[(size_t)t * nb + n]; b += s[(size_t)t * nb + n]; }
    cbar[n] = a / L;
    sbar[n] = b / L;
}

// block = 64 sites x 4 tau-phases; each thread sums every 4th slice with 4 independent accumulators
// (loads in flight instead of one dependent chain), phases combined through LDS.
__global__ void __launch_bounds__(256) k_ebar(double *__restrict__ Ebar, const double *__restrict__ E, int N, int L) {
    __shared__ double part[4][64];
    Ebar += (size_t)blockIdx.y * N;                    // blockIdx.y = chain
    E += (size_t)blockIdx.y * (size_t)N * L;
    const int lane = threadIdx.x & 63, ph = threadIdx.x >> 6;
    const int s = blockIdx.x * 64 + lane;
    const int sc = (s < N) ? s : N - 1;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int t = ph;
    for (; t + 12 < L; t += 16) {
        a0 += E[(size_t)t * N + sc];
        a1 += E[(size_t)(t + 4) * N + sc];
        a2 += E[(size_t)(t + 8) * N + sc];
        a3 += E[(size_t)(t + 12) * N + sc];
    }
    for (; t < L; t += 4) a0 += E[(size_t)t * N + sc];
    part[ph][lane] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (ph == 0 && s < N) Ebar[s] = ((part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane])) / (double)L;
}

__global__ void k_copy(double *__restrict__ dst, const double *__restrict__ src, long long n) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

// partial r.z when the preconditioner is inactive (identity copy, KPMPreconditioners.jl:475-478)
__global__ void __launch_bounds__(WAVE) k_copy_dot(double *__restrict__ zp, const double *__restrict__ r,
                                                   double *__restrict__ rz_part, int nrz, int N, int L,
                                                   const CgState *state) {
    const int t = blockIdx.x, rhs = blockIdx.y;
    if (state && state[2 * rhs].done) return;   // `state` points at the current copy
    double a = 0.0;
    for (int s = threadIdx.x; s < N; s += WAVE) {
        const size_t i = (size_t)rhs * N * L + (size_t)t * N + s;
        const double v = r[i];
        zp[i] = v;
        a += v * v;
    }
    a = wave_sum(a);
    if (threadIdx.x == 0) {
        // slot t gets the slice sum; this block also clears its share of the unused slots [L, nrz)
        rz_part[(size_t)rhs * nrz + t] = a;
        for (int q = L + t; q < nrz; q += L) rz_part[(size_t)rhs * nrz + q] = 0.0;
    }
}

// ------------------------------------------------------------------------------------------
// Fermion force of the Holstein model: the callers' work around the two solves.
//   k_lambda_rhs   b = Lambda phi            update_Λ! + mulΛ!, HMC.jl:921-968
//   k_force_holstein  dSf/dx                  calc_dSfdx! = mulM! + muldMdx! + muldΛdx!,
//                                             HMC.jl:790-814, HolsteinModels.jl:691-755, HMC.jl:1005-1025
// Everything a slice needs lives at tau and tau-1, so the force is one pass: for both pseudofermion fields at
// once (two LDS slabs, one sweep),  u = (M X)(tau) = X(tau) - sg CB[E(tau) X(tau-1)],  y = CB^T u,
//   F(tau) = sum_± [ -y sg dtau (lambda + 2 lambda2 x) E(tau) X(tau-1)  +  phi(tau) sg dtau (lambda/2 + lambda2 x) Lambda(tau) X(tau-1) ].
// ------------------------------------------------------------------------------------------

__global__ void __launch_bounds__(WAVE) k_lambda_rhs(double *__restrict__ b, const double *__restrict__ phi,
                                                     const double *__restrict__ xS, const double *__restrict__ lam3,
                                                     int N, int L, double dtau) {
    // vectors are laid out [sign v][chain c][ndim]; x is [chain][ndim]   (one chain: gridDim.z == 1)
    const int t = blockIdx.x, v = blockIdx.y, c = blockIdx.z, nch = gridDim.z;
    const size_t ndim = (size_t)N * L, vo = ((size_t)v * nch + c) * ndim;
    const int tp1 = (t == L - 1) ? 0 : t + 1;
    const double sg = (t == L - 1) ? 1.0 : -1.0;            // (Λϕ)[τ] = -Λ[τ+1]ϕ[τ+1];  (Λϕ)[Lτ] = +Λ[1]ϕ[1]
    for (int s = threadIdx.x; s < N; s += WAVE) {
        const double x = xS[(size_t)c * ndim + (size_t)tp1 * N + s];
        const double Lam = exp(-dtau * (lam3[s] * x + lam3[N + s] * (x * x)) / 2);
        b[vo + (size_t)t * N + s] = sg * Lam * phi[vo + (size_t)tp1 * N + s];
    }
}

template <int NPL>
__global__ void __launch_bounds__(1024) k_force_holstein(double *__restrict__ F, const double *__restrict__ X,
                                                         const double *__restrict__ phi, const double *__restrict__ xS,
                                                         const double *__restrict__ lam3, ModelDev m, double dtau) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *bufA = lds, *bufB = lds + m.N;
    const int N = m.N, L = m.L;
    const size_t ndim = (size_t)N * L;
    const int t = blockIdx.x, ch = blockIdx.y, nch = gridDim.y;          // X, phi: [sign][chain][ndim]; F, x: [chain][ndim]
    const size_t minus = (size_t)nch * ndim;                             // offset of the "-" pseudofermion block
    X += (size_t)ch * ndim; phi += (size_t)ch * ndim; xS += (size_t)ch * ndim; F += (size_t)ch * ndim;
    const int tm1 = (t == 0) ? L - 1 : t - 1;
    const double sg = (t == 0) ? -1.0 : 1.0;
    const double *c0 = m.c + (size_t)t * m.cs_tau_stride, *s0 = m.s + (size_t)t * m.cs_tau_stride;
    const double *E0 = m.E + (size_t)(ch % m.nchains) * m.E_chain_stride + (size_t)t * m.E_tau_stride;
    double xmp[NPL], xmm[NPL], x0p[NPL], x0m[NPL], e[NPL];
#pragma unroll
    for (int q = 0; q < NPL; ++q) {
        const int s = threadIdx.x + q * blockDim.x;
        if (s < N) {
            xmp[q] = X[(size_t)tm1 * N + s]; xmm[q] = X[minus + (size_t)tm1 * N + s];
            x0p[q] = X[(size_t)t * N + s]; x0m[q] = X[minus + (size_t)t * N + s];
            e[q] = E0[s];
            bufA[s] = e[q] * xmp[q];
            bufB[s] = e[q] * xmm[q];
        }
    }
    __syncthreads();
    cb_sweep<2, false>(bufA, bufB, c0, s0, c0, s0, m);
#pragma unroll
    for (int q = 0; q < NPL; ++q) {
        const int s = threadIdx.x + q * blockDim.x;
        if (s < N) {
            bufA[s] = x0p[q] - sg * bufA[s];          // (M X+)(tau)
            bufB[s] = x0m[q] - sg * bufB[s];
        }
    }
    __syncthreads();
    cb_sweep<2, true>(bufA, bufB, c0, s0, c0, s0, m);
#pragma unroll
    for (int q = 0; q < NPL; ++q) {
        const int s = threadIdx.x + q * blockDim.x;
        if (s < N) {
            const size_t i = (size_t)t * N + s;
            const double x = xS[i], lam = lam3[s], lam2 = lam3[N + s];
            const double Lam = exp(-dtau * (lam * x + lam2 * (x * x)) / 2);
            const double gM = sg * dtau * (lam + 2 * lam2 * x) * e[q];        // muldMdx! (HolsteinModels.jl:727-741)
            const double gL = sg * dtau * (lam / 2 + lam2 * x) * Lam;         // muldΛdx! (HMC.jl:1015-1022)
            F[i] = -(bufA[s] * (gM * xmp[q])) - (bufB[s] * (gM * xmm[q])) + phi[i] * (gL * xmp[q]) + phi[minus + i] * (gL * xmm[q]);
        }
    }
}

// muldMdx!(dMdx, u, holstein, v) for given u, v (HolsteinModels.jl:691-755) — the Langevin force -2 gᵀ(∂M/∂x)M⁻¹g
// (LangevinDynamics.jl:350-384):  F(tau) = scale * [CBᵀ u](tau) .* sg(tau) dtau (lambda + 2 lambda2 x) E(tau) .* v(tau-1)
template <int NPL>
__global__ void __launch_bounds__(1024) k_dmdx_holstein(double *__restrict__ F, const double *__restrict__ u,
                                                        const double *__restrict__ v, const double *__restrict__ xS,
                                                        const double *__restrict__ lam3, ModelDev m, double dtau, double scale) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *buf = lds;
    const int N = m.N, L = m.L;
    const int t = blockIdx.x, ch = blockIdx.y;                            // blockIdx.y = chain: all vectors are [chain][ndim]
    const size_t co = (size_t)ch * (size_t)N * L;
    F += co; u += co; v += co; xS += co;
    const int tm1 = (t == 0) ? L - 1 : t - 1;
    const double sg = (t == 0) ? -1.0 : 1.0;
    const double *c0 = m.c + (size_t)t * m.cs_tau_stride, *s0 = m.s + (size_t)t * m.cs_tau_stride;
    const double *E0 = m.E + (size_t)(ch % m.nchains) * m.E_chain_stride + (size_t)t * m.E_tau_stride;
#pragma unroll
    for (int q = 0; q < NPL; ++q) {
        const int s = threadIdx.x + q * blockDim.x;
        if (s < N) buf[s] = u[(size_t)t * N + s];
    }
    __syncthreads();
    cb_sweep<1, true>(buf, nullptr, c0, s0, nullptr, nullptr, m);
#pragma unroll
    for (int q = 0; q < NPL; ++q) {
        const int s = threadIdx.x + q * blockDim.x;
        if (s < N) {
            const size_t i = (size_t)t * N + s;
            const double x = xS[i];
            F[i] = scale * buf[s] * (sg * dtau * (lam3[s] + 2 * lam3[N + s] * x) * E0[s] * v[(size_t)tm1 * N + s]);
        }
    }
}

// Fermion force of the SSH model: muldMdx! on bond phonons (SSHModels.jl:707-829) fused with mulM! (HMC.jl:797-806).
// Per slice tau and per pseudofermion field X:
//   b0 = E_mu .* X(tau-1);  u = X(tau) - sg CB_tau b0;  c0 = CB_tau^T u;
//   then bond by bond in checkerboard order  b <- rot_n b,  c <- rot_n^-1 c,  q[tau][n] = c_j b_i + c_i b_j
// (bonds of one colour are site-disjoint => one colour = one parallel step).  The kernel returns
// q = q(X+) + q(X-); the caller multiplies by sg(tau) dtau dK_n/dx and scatters to the phonon fields.
template <int NPL>
// U (optional): muldMdx!(·, u, ssh, v) for a GIVEN u (Langevin: u = g, v = X = M⁻¹g, one field): c₀ = CBᵀ U instead of CBᵀ(M X),
// the second slab carries zeros.
__global__ void __launch_bounds__(1024) k_force_ssh(double *__restrict__ q, const double *__restrict__ X, ModelDev m,
                                                    const double *__restrict__ U, int nch) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int N = m.N, L = m.L;
    double *bP = lds, *cP = lds + N, *bM = lds + 2 * N, *cM = lds + 3 * N;
    const size_t ndim = (size_t)N * L;
    const int t = blockIdx.x;
    // blockIdx.y = chain: X = [sign][chain][ndim] (X+ of chain c, then X- at +nch*ndim), q = [chain][tau][bond], U = [chain][ndim]
    const size_t xm_off = (size_t)nch * ndim;
    { const int ch = blockIdx.y; ssh_chain_select(m, ch); X += (size_t)ch * ndim; q += (size_t)ch * L * m.nb; if (U) U += (size_t)ch * ndim;
      m.E += (size_t)(ch % m.nchains) * (size_t)m.E_chain_stride; }
    const int tm1 = (t == 0) ? L - 1 : t - 1;
    const double sg = (t == 0) ? -1.0 : 1.0;
    const double *ct = m.c + (size_t)t * m.cs_tau_stride, *st = m.s + (size_t)t * m.cs_tau_stride;
    double x0p[NPL], x0m[NPL], b0p[NPL], b0m[NPL];
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
        const int s = threadIdx.x + k * blockDim.x;
        if (s < N) {
            const double e = m.E[s];
            x0p[k] = X[(size_t)t * N + s]; x0m[k] = U ? 0.0 : X[xm_off + (size_t)t * N + s];
            b0p[k] = e * X[(size_t)tm1 * N + s]; b0m[k] = U ? 0.0 : e * X[xm_off + (size_t)tm1 * N + s];
            cP[s] = b0p[k]; cM[s] = b0m[k];
        }
    }
    __syncthreads();
    cb_sweep<2, false>(cP, cM, ct, st, ct, st, m);                      // CB_tau b0
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
        const int s = threadIdx.x + k * blockDim.x;
        if (s < N) {
            cP[s] = U ? U[(size_t)t * N + s] : x0p[k] - sg * cP[s];      // u = (M X)(tau), or the caller's u
            cM[s] = x0m[k] - sg * cM[s];
            bP[s] = b0p[k]; bM[s] = b0m[k];
        }
    }
    __syncthreads();
    cb_sweep<2, true>(cP, cM, ct, st, ct, st, m);                       // c0 = CB_tau^T u
    for (int col = 0; col < m.ncol; ++col) {
        const int n0 = m.coloff[col], n1 = m.coloff[col + 1];
        for (int n = n0 + threadIdx.x; n < n1; n += blockDim.x) {
            const int i = m.bi[n], j = m.bj[n];
            const double cn = ct[n], sn = st[n];
            double acc = 0.0;
            {
                const double bi = bP[i], bj = bP[j], ci = cP[i], cj = cP[j];
                const double nbi = cn * bi + sn * bj, nbj = cn * bj + sn * bi;
                const double nci = cn * ci - sn * cj, ncj = cn * cj - sn * ci;
                bP[i] = nbi; bP[j] = nbj; cP[i] = nci; cP[j] = ncj;
                acc += ncj * nbi + nci * nbj;
            }
            {
                const double bi = bM[i], bj = bM[j], ci = cM[i], cj = cM[j];
                const double nbi = cn * bi + sn * bj, nbj = cn * bj + sn * bi;
                const double nci = cn * ci - sn * cj, ncj = cn * cj - sn * ci;
                bM[i] = nbi; bM[j] = nbj; cM[i] = nci; cM[j] = ncj;
                acc += ncj * nbi + nci * nbj;
            }
            q[(size_t)t * m.nb + n] = acc;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// host-side launchers
// ------------------------------------------------------------------------------------------

ModelDev elph_model_dev(const elph_handle_s *h) {
    ModelDev m;
    m.N = (int)h->N; m.L = (int)h->L; m.nb = (int)h->nb; m.ncol = h->ncol;
    { static const bool nosweep = []() { const char *e = getenv("ELPH_DBG_NOSWEEP"); return e && e[0] == '1'; }(); if (nosweep) m.ncol = 0; }   // timing experiments only
    m.cs_tau_stride = (h->kind == ELPH_MODEL_SSH) ? (int)h->nb : 0;
    m.E_tau_stride = (h->kind == ELPH_MODEL_SSH) ? 0 : (int)h->N;
    m.nchains = h->nchains;
    m.E_chain_stride = (h->kind == ELPH_MODEL_SSH) ? (long long)h->N : (long long)h->ndim;   // SSH: exp(dtau mu), one per-site vector per chain
    m.bi = h->d_bi; m.bj = h->d_bj; m.coloff = h->d_coloff;
    m.c = h->d_c; m.s = h->d_s; m.E = h->d_E;
    if (h->solo_chain >= 0) {   // one right-hand side of a chains batch re-solved alone: present ITS configuration as the only one
        m.nchains = 1;
        m.E = h->d_E + (size_t)h->solo_chain * (size_t)h->ndim;
    }
    m.lp_ij = h->d_lp_ij; m.lp_c = h->d_lp_c; m.lp_s = h->d_lp_s;
    m.lp_tau_stride = (h->kind == ELPH_MODEL_SSH) ? h->lp_ne * ELPH_WAVE : 0;
    m.cs_chain_stride = m.lp_chain_stride = 0;
    m.uniform = 0; m.c_uni = 1.0; m.s_uni = 0.0;
    m.sq_bond = (h->sq_L > 0) ? h->d_sq_bond : nullptr;
    m.grid_GX = h->sq_LX / 2; m.grid_GY = h->sq_LY / 2;
    if (h->pg_kind == 3 && h->pg_L <= 16) m.grid_GX = m.grid_GY = h->pg_L / 2;      // an even-L triangular lattice: the same grid of 2 x 2 patches (cg_wg.hip: FORM 7)
    m.hc_LX = h->hc_LX; m.hc_LY = h->hc_LY;
    if (h->kind == ELPH_MODEL_HOLSTEIN && h->nb > 0) {
        bool uni = true;
        for (int64_t n = 1; n < h->nb && uni; ++n) uni = (h->h_c[(size_t)n] == h->h_c[0] && h->h_s[(size_t)n] == h->h_s[0]);
        if (uni) { m.uniform = 1; m.c_uni = h->h_c[0]; m.s_uni = h->h_s[0]; }
    }
    if (h->kind == ELPH_MODEL_SSH && h->nchains > 1) {
        const long long cs = (long long)h->L * h->nb, lp = (long long)h->L * h->lp_ne * ELPH_WAVE;
        if (h->solo_chain >= 0) {
            m.c += (size_t)h->solo_chain * cs; m.s += (size_t)h->solo_chain * cs;
            m.lp_c += (size_t)h->solo_chain * lp; m.lp_s += (size_t)h->solo_chain * lp;
            m.E = h->d_E + (size_t)h->solo_chain * (size_t)h->N;
        } else {
            m.cs_chain_stride = cs; m.lp_chain_stride = lp;
        }
    }
    return m;
}

KpmDev elph_kpm_dev(const elph_handle_s *h) {
    KpmDev K;
    K.active = h->kpm_active;
    K.Lo2 = (int)((h->L + 1) / 2);
    K.lam_avg = h->lam_avg; K.lam_mag = h->lam_mag;
    K.nchains = h->kpm_nch; K.lam = h->d_klam;
    K.Ebar = h->d_Ebar; K.cbar = h->d_cbar; K.sbar = h->d_sbar;
    K.order = h->d_order; K.coff = h->d_coff; K.coeff = h->d_coeff; K.wsched = h->d_wsched; K.desc = h->d_kdesc;
    K.lp_cbar = h->d_lp_cbar; K.lp_sbar = h->d_lp_sbar;
    const bool hop_per_chain = (h->kind == ELPH_MODEL_SSH && h->kpm_nch > 1);
    K.hop_stride = hop_per_chain ? (long long)h->nb : 0;
    K.lp_hop_stride = hop_per_chain ? (long long)h->lp_ne * ELPH_WAVE : 0;
    K.sq_stride = hop_per_chain ? 4LL * h->N : 0;
    if (h->solo_chain >= 0 && h->kpm_nch > 1) {
        const int c = h->solo_chain;
        K.cbar += (size_t)c * K.hop_stride; K.sbar += (size_t)c * K.hop_stride;
        K.lp_cbar += (size_t)c * K.lp_hop_stride; K.lp_sbar += (size_t)c * K.lp_hop_stride;
        K.nchains = 1;
        K.lam_avg = h->h_lam[2 * c]; K.lam_mag = h->h_lam[2 * c + 1];
        K.Ebar += (size_t)c * h->N;
        K.order += (size_t)c * K.Lo2; K.coff += (size_t)c * (K.Lo2 + 1); K.wsched += (size_t)c * K.Lo2; K.desc += (size_t)c * K.Lo2;
    }
    return K;
}

// generic kernels: workgroup of BS = 64*W threads, W = ceil(N/512); NPL = ceil(N/BS) <= 8
// block size of the generic (LDS-slab) kernels: one wave up to 512 sites; beyond that one thread per two sites (= per bond of a
// full colour), so that a colour sweep is one table read + one LDS round trip per thread instead of four in sequence
static inline int gen_bs(const elph_handle_s *h) {
    if (h->N <= 512) return ELPH_WAVE;
    return std::min(1024, ELPH_WAVE * (int)((h->N / 2 + ELPH_WAVE - 1) / ELPH_WAVE));
}
static inline int gen_npl(const elph_handle_s *h) { const int bs = gen_bs(h); return (int)((h->N + bs - 1) / bs); }

#define DISPATCH_NPL(npl, CALL)                                   \
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

static int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        elph_set_error("launch %s failed: %s", what, hipGetErrorString(e));
        return ELPH_E_HIP;
    }
    return ELPH_OK;
}

// ncols: columns of the vectors (0 = the lattice sites; SSH phonon fields have Nph columns)
int elph_launch_r2s(elph_handle_s *h, double *dstS, const double *srcR, int nvec, int ncols) {
    // in: rows = N sites, cols = L
    const int N = ncols > 0 ? ncols : (int)h->N;
    dim3 grid((unsigned)((h->L + 31) / 32), (unsigned)((N + 31) / 32), (unsigned)nvec);
    hipLaunchKernelGGL(k_transpose, grid, dim3(32, 8), 0, h->stream, dstS, srcR, N, (int)h->L);
    return check_launch("k_transpose(r2s)");
}

int elph_launch_s2r(elph_handle_s *h, double *dstR, const double *srcS, int nvec, int ncols) {
    // in: rows = L, cols = N
    const int N = ncols > 0 ? ncols : (int)h->N;
    dim3 grid((unsigned)((N + 31) / 32), (unsigned)((h->L + 31) / 32), (unsigned)nvec);
    hipLaunchKernelGGL(k_transpose, grid, dim3(32, 8), 0, h->stream, dstR, srcS, (int)h->L, N);
    return check_launch("k_transpose(s2r)");
}

int elph_launch_expV(elph_handle_s *h, const double *xR, double dtau, int chain) {
    dim3 grid((unsigned)((h->L + 31) / 32), (unsigned)((h->N + 31) / 32), 1);
    hipLaunchKernelGGL(k_expV, grid, dim3(32, 8), 0, h->stream, h->d_E + (size_t)chain * (size_t)h->ndim, xR, h->d_lam, (int)h->N, (int)h->L, dtau);
    return check_launch("k_expV");
}

int elph_launch_zero(elph_handle_s *h, double *p, int64_t n) {
    if (n <= 0) return ELPH_OK;
    hipLaunchKernelGGL(k_zero, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, p, (long long)n);
    return check_launch("k_zero");
}

int elph_launch_mul(elph_handle_s *h, int which, double *yS, const double *vS, int nvec) {
    if (h->fast) return elph_fast_mul(h, which, yS, vS, nvec);
    ModelDev m = elph_model_dev(h);
    if (which >= 0 && which <= 2 && elph_pg_mul_usable(h) && (m.uniform || elph_pg_disorder_ok(h))) return elph_pg_mul(h, m, which, yS, vS, nvec);     // large square / honeycomb lattices: pgrid.hip
    dim3 grid((unsigned)h->L, (unsigned)nvec, 1);
    const size_t shm = (2 * (size_t)h->N + 16) * sizeof(double);
    DISPATCH_NPL(gen_npl(h), {
        if (which == 0) hipLaunchKernelGGL((k_mul<NPL, 0>), grid, dim3((unsigned)gen_bs(h)), shm, h->stream, yS, vS, m);
        else if (which == 1) hipLaunchKernelGGL((k_mul<NPL, 1>), grid, dim3((unsigned)gen_bs(h)), shm, h->stream, yS, vS, m);
        else hipLaunchKernelGGL((k_mul<NPL, 2>), grid, dim3((unsigned)gen_bs(h)), shm, h->stream, yS, vS, m);
    });
    return check_launch("k_mul");
}

static CgBufs make_bufs(elph_handle_s *h, int nrhs) {
    CgBufs B;
    const size_t P = (size_t)h->cap_rhs * (size_t)h->L * (size_t)(h->npl);  // partial stride per array
    B.x = h->d_x; B.r = h->d_r; B.z = h->d_z; B.zp = h->d_zp; B.p = h->d_p;
    B.pap = h->d_part; B.rr = h->d_part + P; B.rz = h->d_part + 2 * P;
    B.state = h->d_state; B.params = h->cur_params; B.hist = h->d_hist;
    B.alpha = h->d_alpha;
    B.dot_lo = h->dot_hi > 0 ? h->dot_lo : 0;
    B.dot_hi = h->dot_hi > 0 ? h->dot_hi : (int)h->N;
    B.nrz = (int)(h->L * h->npl);
    {   // slices per wave of k_cg_ap: the p/x-fused kernel of a preconditioned batch has its own rule
        const int Tc = (h->px_solve && h->cur_params.use_prec && h->px_via_pg) ? 1      // (k_cg_ap_pg: one p.z slot per time slice)
                       : (h->px_solve && h->cur_params.use_prec) ? elph_choose_T_px(h, h->T_rhs_hint > 0 ? h->T_rhs_hint : nrhs) : elph_choose_T(h, nrhs);
        B.npap = (int)((h->L + Tc - 1) / Tc);      // (a ragged cut: ceil)
    }
    B.nrhs = nrhs;
    return B;
}

int elph_launch_ssh_update(elph_handle_s *h, const double *x_dev, int nph, const int *cb0_dev, const double *par_dev,
                           const double *tbare_dev, const int *slot_dev, double dtau, int x_tau_major, int nch) {
    const int nb = (int)h->nb, L = (int)h->L, lp_stride = h->lp_ne * ELPH_WAVE;
    const int *slot = h->fast_capable ? slot_dev : nullptr;
    if (nch > h->ssh_chain_cap) { elph_set_error("%d chains, tables for %d", nch, h->ssh_chain_cap); return ELPH_E_STATE; }
    if (nb > 0) {
        const long long n1 = (long long)nb * L;
        hipLaunchKernelGGL(k_ssh_fill, dim3((unsigned)((n1 + 255) / 256), (unsigned)nch), dim3(256), 0, h->stream, h->d_c, h->d_s, h->d_lp_c, h->d_lp_s,
                           tbare_dev, slot, nb, L, lp_stride, dtau);
        if (nph > 0) {
            const long long n2 = (long long)nph * L;
            hipLaunchKernelGGL(k_ssh_fields, dim3((unsigned)((n2 + 255) / 256), (unsigned)nch), dim3(256), 0, h->stream, h->d_c, h->d_s, h->d_lp_c,
                               h->d_lp_s, x_dev, par_dev, cb0_dev, slot, nph, nb, L, lp_stride, dtau, x_tau_major);
        }
    }
    {
        const bool per = h->mu_per_chain && h->d_mu_ch && nch <= h->mu_ch_cap;
        hipLaunchKernelGGL(k_ssh_expmu, dim3((unsigned)((h->N + 255) / 256), (unsigned)nch), dim3(256), 0, h->stream, h->d_E,
                           per ? h->d_mu_ch : h->d_lam, (int)h->N, dtau, per ? (int)h->N : 0);
    }
    return check_launch("ssh update_model");
}

int elph_launch_ssh_scatter(elph_handle_s *h, double *F_dev, const double *q_dev, const double *x_dev, const double *par_dev,
                            const int *cb0_dev, int nph, double dtau, int tau_major, double scale, int nch) {
    const long long n = (long long)nph * h->L;
    if (n == 0) return ELPH_OK;
    hipLaunchKernelGGL(k_ssh_scatter, dim3((unsigned)((n + 255) / 256), (unsigned)nch), dim3(256), 0, h->stream, F_dev, q_dev, x_dev, par_dev, cb0_dev,
                       nph, (int)h->nb, (int)h->L, dtau, tau_major, scale);
    return check_launch("k_ssh_scatter");
}

int elph_launch_cs_bar(elph_handle_s *h, double *cbar_dev, double *sbar_dev, int nch) {
    hipLaunchKernelGGL(k_cs_bar, dim3((unsigned)((h->nb + 255) / 256), (unsigned)nch), dim3(256), 0, h->stream, cbar_dev, sbar_dev, h->d_c, h->d_s,
                       (int)h->nb, (int)h->L);
    return check_launch("k_cs_bar");
}

// Ē of the first `nch` resident chains in one launch
int elph_launch_ebar(elph_handle_s *h, int nch) {
    hipLaunchKernelGGL(k_ebar, dim3((unsigned)((h->N + 63) / 64), (unsigned)nch), dim3(256), 0, h->stream, h->d_Ebar, h->d_E,
                       (int)h->N, (int)h->L);
    return check_launch("k_ebar");
}

// Does the preconditioned iteration of THIS solve run p/x-fused (PxFuse, dft_mfma.hip)?  Decided once per solve (elph_launch_cg_init) so
// that k_cg_ap_chunk<PX> and the inverse transform agree: the batched Holstein iteration on a four-colour lane program whose
// Chebyshev kernel leaves r.z in frequency space (so that beta is known BEFORE the inverse transform), residual update folded into the
// forward transform, streaming MFMA inverse with one row group, the templated chunk lengths.  ELPH_FUSE_PX=0: off (A/B, parity tests).
static bool reg_cheb_form(const elph_handle_s *h) {      // a register-exchange Chebyshev kernel (the forms that deliver r.z in frequency space)
    const char *e = getenv("ELPH_NO_SQ");
    if (e && e[0] == '1') return false;
    return h->sq_P > 0 || (h->sq_L > 0 && h->sq_uniform && h->kind == ELPH_MODEL_HOLSTEIN) ||
           (h->hc_L > 0 && h->hc_uniform && h->kind == ELPH_MODEL_HOLSTEIN && (h->hc12 || h->hc_L * h->hc_L <= 64 || (h->hc_L % 2 == 0 && h->hc_L <= 16)));
}

// the patch-form lattices of the generic family (round 6): Holstein, uniform hopping, k_cg_ap_pg + k_kpm_cheb_pg (pgrid.hip) — their p/x-fused
// iteration is the lane-program family's with those two kernels in its place (ELPH_PG_PX=0: the unfused form, A/B; read per call)
static bool pg_px_allowed() {
    const char *e = getenv("ELPH_PG_PX");
    return !(e && e[0] == '0');
}
static bool pg_px_form(const elph_handle_s *h) {
    if (!pg_px_allowed()) return false;
    return !h->fast && h->kind == ELPH_MODEL_HOLSTEIN && (h->pg_uniform || elph_pg_disorder_ok(h)) && elph_pg_ap_usable(h) && elph_pg_cheb_usable(h);
}

static bool px_plan(elph_handle_s *h, int nrhs) {
    if (!h->kpm_active) return false;
    if (pg_px_form(h)) {
        const int N = (int)h->N, L = (int)h->L, Lo2 = (L + 1) / 2;
        CgBufs B = make_bufs(h, nrhs);
        if (!(B.dot_lo == 0 && B.dot_hi == N) || B.npap != L || 2 * Lo2 > B.nrz) return false;
        const char *ef = getenv("ELPH_FREQ_RZ");
        if (ef && ef[0] == '0') return false;
        return elph_dft_mfma_xr_usable(h, N, nrhs) && elph_dft_mfma_px_usable(h, N, nrhs);
    }
    if (!h->fast) {
        // (round 6) every other lattice of the generic family — ragged colours, more than six colours, hopping disorder, no patch form, bond
        // phonons beyond the lane-program sizes: k_cg_ap<PX> + k_kpm_cheb with r.z in frequency space (ELPH_GEN_PX=0: the unfused form, A/B)
        const char *eg = getenv("ELPH_GEN_PX");
        if (eg && eg[0] == '0') return false;
        const int N = (int)h->N, L = (int)h->L, Lo2 = (L + 1) / 2;
        CgBufs B = make_bufs(h, nrhs);
        if (!(B.dot_lo == 0 && B.dot_hi == N) || B.npap != L || Lo2 > B.nrz || elph_pg_cheb_usable(h)) return false;
        const char *ef = getenv("ELPH_FREQ_RZ");
        if (ef && ef[0] == '0') return false;
        return elph_dft_mfma_xr_usable(h, N, nrhs) && elph_dft_mfma_px_usable(h, N, nrhs);
    }
    if (h->lp_mc != 4) {
        // six-colour lane programs (triangular lattices up to 16 x 16: the geometry of holstein_hmc_triangular.toml) have no fused chunk kernel of
        // their own: their p/x-fused iteration takes the patch-form pair k_cg_ap_pg<PX> + k_kpm_cheb_pg (pgrid::Tri<2, 2>) — round 6
        if (!(pg_px_allowed() && h->kind == ELPH_MODEL_HOLSTEIN && h->pg_L > 0 && h->pg_uniform && elph_pg_cheb_usable(h))) return false;
        const int N = (int)h->N, L = (int)h->L, Lo2 = (L + 1) / 2;
        const bool keep = h->px_via_pg, keeps = h->px_solve;
        h->px_via_pg = true; h->px_solve = true;            // (make_bufs prices the partial-sum layout of the form being planned)
        CgParams kp = h->cur_params; h->cur_params.use_prec = 1;
        CgBufs B = make_bufs(h, nrhs);
        h->cur_params = kp; h->px_via_pg = keep; h->px_solve = keeps;
        if (!(B.dot_lo == 0 && B.dot_hi == N) || B.npap != L || 2 * Lo2 > B.nrz) return false;
        const char *ef = getenv("ELPH_FREQ_RZ");
        if (ef && ef[0] == '0') return false;
        return elph_dft_mfma_xr_usable(h, N, nrhs) && elph_dft_mfma_px_usable(h, N, nrhs);
    }
    const int N = (int)h->N, L = (int)h->L, Lo2 = (L + 1) / 2;
    CgBufs B = make_bufs(h, nrhs);
    if (!(B.dot_lo == 0 && B.dot_hi == N) || !elph_dft_mfma_xr_usable(h, N, nrhs)) return false;      // the iteration takes cg_mode 2
    const char *ef = getenv("ELPH_FREQ_RZ");
    if ((ef && ef[0] == '0') || 2 * Lo2 > B.nrz) return false;
    // (r.z in frequency space comes from a register-exchange Chebyshev kernel — or, round 6, from the patch-form one: square L = 18, 20 of this family)
    // or from the Re / Im recursion through the LDS slab (k_kpm_cheb_ri: square L = 22, disordered honeycomb lattices, ...; ELPH_LDS_CHEB_PX=0 and
    // ELPH_NO_SQ=1 — the A/B that forces that recursion on a lattice with a register form — keep the unfused iteration)
    if (!reg_cheb_form(h) && !(elph_pg_cheb_usable(h) && pg_px_allowed())) {
        const char *el = getenv("ELPH_LDS_CHEB_PX"), *ens = getenv("ELPH_NO_SQ");
        if ((el && el[0] == '0') || (ens && ens[0] == '1') || elph_pg_cheb_usable(h)) return false;
    }
    { const char *ec = getenv("ELPH_CHEB_COMPLEX"); if (ec && ec[0] == '1') return false; }
    const int T = elph_choose_T_px(h, h->T_rhs_hint > 0 ? h->T_rhs_hint : nrhs);
    if (!(T > 1 && L % T == 0 && (T == 20 || T == 16 || T == 10 || T == 8 || T == 5 || T == 4 || T == 2))) return false;
    return elph_dft_mfma_px_usable(h, N, nrhs);
}

bool elph_px_plan(elph_handle_s *h, int nrhs) { return px_plan(h, nrhs); }

// z = P^-1 r on layout-S vectors.  cg_mode: 0 standalone; 1 inside CG (skip when done, fuse r.z partials); 2 as 1 with the
// residual update r -= alpha A p (k_cg_xr) folded into the forward transform (rS is then written)
// parts (measurement only, bench.py's per-kernel times): bit 0 forward transform, bit 1 Chebyshev recursion, bit 2 inverse transform
int elph_launch_kpm_apply(elph_handle_s *h, double *zS, const double *rS, int nrhs, int cg_mode, int parts) {
    const int N = (int)h->N, L = (int)h->L, Lo2 = (L + 1) / 2;
    CgBufs B = make_bufs(h, nrhs);
    // kernels that skip finished right-hand sides read the state copy written by the latest k_cg_ap launch
    const CgState *st = cg_mode ? h->d_state + (h->ap_count & 1) : nullptr;
    if (!h->kpm_active) {
        if (cg_mode) {
            hipLaunchKernelGGL(k_copy_dot, dim3((unsigned)L, (unsigned)nrhs), dim3(WAVE), 0, h->stream, zS, rS, B.rz,
                               B.nrz, N, L, st);
        } else {
            const long long n = (long long)nrhs * N * L;
            hipLaunchKernelGGL(k_copy, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, zS, rS, n);
        }
        return check_launch("kpm identity");
    }
    KpmDev K = elph_kpm_dev(h);
    ModelDev m = elph_model_dev(h);

    // FOLD (dft_mfma.hip: XrFuse): in the batched CG iteration (forward transform with the residual update folded in, register-exchange
    // Chebyshev kernel with the r.z partials in frequency space) the frequencies of order 1 — z_w = |c0|^2 r_w — are finished by the
    // forward transform; the Chebyshev kernel keeps their slot bookkeeping only
    static const bool freq_rz_on = []() { const char *e = getenv("ELPH_FREQ_RZ"); return !(e && e[0] == '0'); }();
    const int nct = (N + 15) / 16;
    const bool no_sq = []() { const char *e = getenv("ELPH_NO_SQ"); return e && e[0] == '1'; }();                    // (A/B: the LDS recursion, which knows no fold)
    const bool reg_cheb = !no_sq && (h->sq_P > 0 || (h->sq_L > 0 && h->sq_uniform && h->kind == ELPH_MODEL_HOLSTEIN) || (h->hc_L > 0 && h->hc_uniform && h->kind == ELPH_MODEL_HOLSTEIN && (h->hc12 || h->hc_L * h->hc_L <= 64 || (h->hc_L % 2 == 0 && h->hc_L <= 16))));     // (a register-exchange Chebyshev kernel: the one that knows the fold)
    const bool fold = cg_mode == 2 && h->fast && reg_cheb && h->lp_mc == 4 && freq_rz_on && h->d_kfold &&
                      2 * Lo2 + nct <= B.nrz && B.dot_lo == 0 && B.dot_hi == N && elph_dft_mfma_fold_usable(h);
    if (!(parts & 1)) {
    } else if (cg_mode == 2) {
        int rcd = elph_dft_mfma_fwd_xr(h, h->d_nu, const_cast<double *>(rS), B.z, B.pap, B.npap, B.rr, B.alpha, N, nrhs, st,
                                       fold ? h->d_kfold + (h->solo_chain >= 0 && h->kpm_nch > 1 ? 2 * (size_t)h->solo_chain * Lo2 : 0) : nullptr,
                                       (h->solo_chain >= 0) ? 1 : std::max(1, h->kpm_nch), B.rz, B.nrz, 2 * Lo2);
        if (rcd) return rcd;
    } else {
        int rcd = elph_dft_fwd_twisted(h, h->d_nu, rS, N, nrhs, st);
        if (rcd) return rcd;
    }
    const size_t shm = (size_t)N * sizeof(double2);
    bool rz_done = false;     // r.z partials already produced in frequency space by the Chebyshev kernel
    if (!(parts & 2)) {
        rz_done = h->fast && reg_cheb && cg_mode;      // (timing the inverse transform alone: the form that follows the register-exchange kernel)
    } else if (elph_pg_cheb_usable(h)) {
        // an even-L square lattice beyond 16 x 16 (L = 18 ... 32), uniform hopping: the recursion in registers, a patch of sites per lane
        // (whether the lattice still fits the lane-program family — 18 x 18, 20 x 20 — or only the generic kernels)
        static const bool freq_rz = []() { const char *e = getenv("ELPH_FREQ_RZ"); return !(e && e[0] == '0'); }();
        const bool want = cg_mode && freq_rz && 2 * Lo2 <= B.nrz && B.dot_lo == 0 && B.dot_hi == N && h->px_solve;      // (r.z in frequency space: the p/x-fused iteration's)
        int rcp = elph_pg_kpm_cheb(h, nrhs, st, want ? B.rz : nullptr, B.nrz, B.rr);
        if (rcp) return rcp;
        rz_done = want;
    } else if (h->fast) {
        static const bool freq_rz = []() { const char *e = getenv("ELPH_FREQ_RZ"); return !(e && e[0] == '0'); }();
        const bool want = cg_mode && freq_rz && 2 * Lo2 <= B.nrz && B.dot_lo == 0 && B.dot_hi == N;
        int rcf = elph_fast_kpm_cheb(h, nrhs, st, want ? B.rz : nullptr, B.nrz, &rz_done, B.rr, (fold && want) ? nct : 0);
        if (rcf) return rcf;
    } else {
        static const bool gfreq_rz = []() { const char *e = getenv("ELPH_FREQ_RZ"); return !(e && e[0] == '0'); }();
        const bool gwant = cg_mode && gfreq_rz && Lo2 <= B.nrz && B.dot_lo == 0 && B.dot_hi == N && h->px_solve;      // (the p/x-fused iteration of the generic family)
        // one thread per bond of the largest colour (up to 1024): a colour is then one LDS round trip per thread; the bond
        // program rides in LDS when it fits next to the slab
        int maxcol = 1;
        for (int cidx = 0; cidx < h->ncol; ++cidx) maxcol = std::max(maxcol, h->h_coloff[(size_t)cidx + 1] - h->h_coloff[(size_t)cidx]);
        const int cbs = std::min(1024, std::max(gen_bs(h), ELPH_WAVE * ((maxcol + ELPH_WAVE - 1) / ELPH_WAVE)));
        const int cnpl = (N + cbs - 1) / cbs;
        const size_t tab = (size_t)h->nb * (2 * sizeof(double) + sizeof(unsigned));
        const int lds_tables = (N <= 65535 && shm + tab <= 64 * 1024) ? 1 : 0;
        DISPATCH_NPL(cnpl, {
            hipLaunchKernelGGL((k_kpm_cheb<NPL>), dim3((unsigned)nrhs, (unsigned)Lo2), dim3((unsigned)cbs), shm + (lds_tables ? tab : 0),
                               h->stream, h->d_nu, K, m, Lo2, st, lds_tables, gwant ? B.rz : nullptr, B.nrz, B.rr);
        });
        rz_done = gwant;
    }
    // r.z partial slots: (blockIdx.y * gridDim.x + blockIdx.x) < ceil(L/TPT)*nst <= L*npl = nrz; the kernel clears the rest
    if ((parts & 4) && cg_mode == 2 && h->px_solve) {
        // p/x-fused tail: x += alpha p, p = P^-1 r + beta p in the epilogue of the inverse transform; P^-1 r itself is not written
        if (!rz_done) { elph_set_error("p/x-fused iteration planned, but the Chebyshev kernel did not deliver r.z (internal error)"); return ELPH_E_STATE; }
        int rcd = elph_dft_mfma_inv_px(h, h->d_nu, N, nrhs, st, h->d_p, h->d_x, B.alpha, B.rz, B.nrz);
        if (rcd) return rcd;
    } else if (parts & 4) {
        const bool fuse = cg_mode && !rz_done;
        int rcd = elph_dft_inv_twisted(h, zS, h->d_nu, N, nrhs, st, fuse ? rS : nullptr, fuse ? B.rz : nullptr, B.nrz);
        if (rcd) return rcd;
    }
    return check_launch("kpm apply");
}

int elph_launch_rz_partials(elph_handle_s *h, int nrhs);

int elph_launch_cg_init(elph_handle_s *h, int nrhs, int use_prec, bool x_zero) {
    // expects d_b (layout S), d_x = initial guess; computes r0, p0 and seeds the state
    CgBufs B = make_bufs(h, nrhs);
    h->ap_count = 0;
    const int N = (int)h->N, L = (int)h->L;
    int rc = ELPH_OK;
    {   // the form of this solve's preconditioned iteration; a captured graph of the other form is dropped
        const bool px = use_prec && h->kpm_ready && px_plan(h, nrhs);
        if (px != h->px_solve) elph_i_drop_graphs(h);
        h->px_solve = px;
        h->px_via_pg = px && h->fast && h->lp_mc != 4;
    }
    h->x_zero_seen = x_zero;
    if (h->x_zero_seen) {       // x0 = 0 (the library zeroed it for this solve): A x0 = 0 without the mat-vec
        if (hipMemsetAsync(h->d_tmp, 0, (size_t)nrhs * (size_t)h->ndim * sizeof(double), h->stream) != hipSuccess) { elph_set_error("memset failed"); return ELPH_E_HIP; }
    } else {
        rc = elph_launch_mul(h, 2, h->d_tmp, h->d_x, nrhs);
        if (rc) return rc;
    }
    const size_t P = (size_t)h->cap_rhs * (size_t)h->L * (size_t)h->npl;
    double *bb = h->d_part + 3 * P;
    hipLaunchKernelGGL(k_cg_init, dim3((unsigned)L, (unsigned)nrhs), dim3(WAVE), 0, h->stream, B, h->d_b, h->d_tmp, bb, N, L);
    rc = check_launch("k_cg_init");
    if (rc) return rc;
    if (use_prec) {
        // z0 = P^-1 r0, p0 = z0, rho0 = r0.z0 (IterativeSolvers.jl:182-189); once per solve
        rc = elph_launch_kpm_apply(h, h->d_zp, h->d_r, nrhs, 0);
        if (rc) return rc;
        rc = elph_launch_rz_partials(h, nrhs);
        if (rc) return rc;
        hipLaunchKernelGGL(k_cg_init_prec, dim3((unsigned)L, (unsigned)nrhs), dim3(WAVE), 0, h->stream, B, N, L);
        rc = check_launch("k_cg_init_prec");
        if (rc) return rc;
    }
    hipLaunchKernelGGL(k_cg_state0, dim3((unsigned)nrhs), dim3(WAVE), 0, h->stream, B, bb, L);
    return check_launch("k_cg_state0");
}

// partial r.zp into B.rz (slot t = slice sum, rest zero)
__global__ void __launch_bounds__(WAVE) k_rz_part(const double *__restrict__ r, const double *__restrict__ zp,
                                                  double *__restrict__ rz_part, int nrz, int N, int L) {
    const int t = blockIdx.x, rhs = blockIdx.y;
    double a = 0.0;
    for (int s = threadIdx.x; s < N; s += WAVE) {
        const size_t i = (size_t)rhs * N * L + (size_t)t * N + s;
        a += r[i] * zp[i];
    }
    a = wave_sum(a);
    if (threadIdx.x == 0) {
        rz_part[(size_t)rhs * nrz + t] = a;
        for (int q = L + t; q < nrz; q += L) rz_part[(size_t)rhs * nrz + q] = 0.0;
    }
}

int elph_launch_rz_partials(elph_handle_s *h, int nrhs) {
    CgBufs B = make_bufs(h, nrhs);
    hipLaunchKernelGGL(k_rz_part, dim3((unsigned)h->L, (unsigned)nrhs), dim3(WAVE), 0, h->stream, h->d_r, h->d_zp, B.rz,
                       B.nrz, (int)h->N, (int)h->L);
    return check_launch("k_rz_part");
}

// one CG iteration's kernels (graph-capturable: no syncs, no allocations, launch-invariant arguments)
int elph_launch_cg_iteration(elph_handle_s *h, int nrhs, int use_prec) {
    CgBufs B = make_bufs(h, nrhs);
    int rc;
    if (h->fast) {
        const bool px = use_prec && h->px_solve;
        if (px && h->px_via_pg) { ModelDev mp = elph_model_dev(h); rc = elph_pg_cg_ap(h, B, mp, nrhs, (int)(h->ap_count & 1), true); }
        else rc = elph_fast_cg_ap(h, B, nrhs, (int)(h->ap_count & 1), px);
        h->ap_count++;
        if (rc) return rc;
        if (use_prec && h->kpm_active && B.dot_lo == 0 && B.dot_hi == (int)h->N && elph_dft_mfma_xr_usable(h, (int)h->N, nrhs))
            return elph_launch_kpm_apply(h, h->d_zp, h->d_r, nrhs, 2);      // k_cg_xr rides on the forward transform
        if (px) { elph_set_error("p/x-fused iteration planned, but the residual update is not folded into the forward transform (internal error)"); return ELPH_E_STATE; }
        rc = elph_fast_cg_xr(h, B, nrhs, (int)(h->ap_count & 1));
        if (rc) return rc;
    } else {
        ModelDev m = elph_model_dev(h);
        const int N = (int)h->N, L = (int)h->L;
        dim3 grid((unsigned)L, (unsigned)nrhs, 1);
        const size_t shm = (2 * (size_t)N + 16) * sizeof(double);
        const bool pg = elph_pg_ap_usable(h) && (m.uniform || elph_pg_disorder_ok(h)) && B.npap == L;      // a large even-L square lattice: the patch-layout kernel (pgrid.hip)
        if (use_prec && h->px_solve) {
            // the p/x-fused iteration of the generic family (px_plan): k_cg_ap_pg (patch-form lattices) or k_cg_ap<PX> reads the ready p; the
            // residual update rides on the forward transform, r.z comes from the Chebyshev kernel in frequency space, the p/x-update is the
            // inverse transform's epilogue
            if (pg) rc = elph_pg_cg_ap(h, B, m, nrhs, (int)(h->ap_count & 1), true);
            else {
                DISPATCH_NPL(gen_npl(h), {
                    hipLaunchKernelGGL((k_cg_ap<NPL, true>), grid, dim3((unsigned)gen_bs(h)), shm, h->stream, B, m, (int)(h->ap_count & 1));
                });
                rc = check_launch("k_cg_ap<PX>");
            }
            h->ap_count++;
            if (rc) return rc;
            return elph_launch_kpm_apply(h, h->d_zp, h->d_r, nrhs, 2);
        }
        if (pg) { rc = elph_pg_cg_ap(h, B, m, nrhs, (int)(h->ap_count & 1)); if (rc) return rc; }
        DISPATCH_NPL(gen_npl(h), {
            if (!pg) hipLaunchKernelGGL((k_cg_ap<NPL>), grid, dim3((unsigned)gen_bs(h)), shm, h->stream, B, m, (int)(h->ap_count & 1));
            hipLaunchKernelGGL((k_cg_xr<NPL>), grid, dim3((unsigned)gen_bs(h)), 0, h->stream, B, N, L, (int)((h->ap_count + 1) & 1));
        });
        h->ap_count++;
        rc = check_launch("cg iteration");
        if (rc) return rc;
    }
    if (use_prec) rc = elph_launch_kpm_apply(h, h->d_zp, h->d_r, nrhs, 1);
    return rc;
}

// one kernel of the iteration alone (measurement only: bench.py times the dominant kernel by itself)
int elph_launch_cg_kernel(elph_handle_s *h, int nrhs, int which) {
    CgBufs B = make_bufs(h, nrhs);
    if (h->fast) {
        if (which == 0) {
            int rc;
            if (B.params.use_prec && h->px_solve && h->px_via_pg) { ModelDev mp = elph_model_dev(h); rc = elph_pg_cg_ap(h, B, mp, nrhs, (int)(h->ap_count & 1), true); }
            else rc = elph_fast_cg_ap(h, B, nrhs, (int)(h->ap_count & 1), B.params.use_prec && h->px_solve);
            h->ap_count++;
            return rc;
        }
        return elph_fast_cg_xr(h, B, nrhs, (int)(h->ap_count & 1));
    }
    ModelDev m = elph_model_dev(h);
    const int N = (int)h->N, L = (int)h->L;
    dim3 grid((unsigned)L, (unsigned)nrhs, 1);
    const size_t shm = (2 * (size_t)N + 16) * sizeof(double);
    if (which == 0 && elph_pg_ap_usable(h) && (m.uniform || elph_pg_disorder_ok(h)) && B.npap == L) {
        int rc = elph_pg_cg_ap(h, B, m, nrhs, (int)(h->ap_count & 1), B.params.use_prec && h->px_solve);
        h->ap_count++;
        return rc;
    }
    const bool gpx = B.params.use_prec && h->px_solve;
    DISPATCH_NPL(gen_npl(h), {
        if (which == 0 && gpx) hipLaunchKernelGGL((k_cg_ap<NPL, true>), grid, dim3((unsigned)gen_bs(h)), shm, h->stream, B, m, (int)(h->ap_count & 1));
        else if (which == 0) hipLaunchKernelGGL((k_cg_ap<NPL>), grid, dim3((unsigned)gen_bs(h)), shm, h->stream, B, m, (int)(h->ap_count & 1));
        else hipLaunchKernelGGL((k_cg_xr<NPL>), grid, dim3((unsigned)gen_bs(h)), 0, h->stream, B, N, L, (int)(h->ap_count & 1));
    });
    if (which == 0) h->ap_count++;
    return check_launch("cg kernel");
}

CgBufs elph_make_bufs(elph_handle_s *h, int nrhs) { return make_bufs(h, nrhs); }

// true residual of d_x against d_b -> d_scal[rhs]
int elph_launch_residual(elph_handle_s *h, int nrhs) {
    int rc = elph_launch_mul(h, 2, h->d_tmp, h->d_x, nrhs);
    if (rc) return rc;
    const size_t P = (size_t)h->cap_rhs * (size_t)h->L * (size_t)h->npl;
    double *pa = h->d_part, *pb = h->d_part + P;
    hipLaunchKernelGGL(k_resid_part, dim3((unsigned)h->L, (unsigned)nrhs), dim3(WAVE), 0, h->stream, h->d_tmp, h->d_b, pa,
                       pb, (int)h->N, (int)h->L);
    hipLaunchKernelGGL(k_resid_final, dim3((unsigned)nrhs), dim3(WAVE), 0, h->stream, pa, pb, h->d_scal, (int)h->L);
    return check_launch("residual");
}

int elph_launch_tau_to_omega(elph_handle_s *h, double2 *nuS_full, const double *vS) {
    const int N = (int)h->N, L = (int)h->L, Lo2 = (L + 1) / 2;

    {
        int rcd = elph_dft_fwd_twisted(h, h->d_nu, vS, N, 1, nullptr);
        if (rcd) return rcd;
    }
    const long long n = (long long)N * L;
    hipLaunchKernelGGL(k_expand_spectrum, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, nuS_full, h->d_nu, N,
                       L, Lo2);
    return check_launch("tau_to_omega");
}

int elph_launch_omega_to_tau(elph_handle_s *h, double *vS, const double2 *nuS_full) {
    const int N = (int)h->N, L = (int)h->L;
    const int nst = (N + WAVE - 1) / WAVE;
    hipLaunchKernelGGL(k_dft_inv_twisted_full, dim3((unsigned)nst, (unsigned)L, 1), dim3(WAVE), 0, h->stream, vS, nuS_full,
                       h->d_theta, N, L);
    return check_launch("omega_to_tau");
}

int elph_launch_fft_accel(elph_handle_s *h, double *outS, const double *inS, const double *diagS, double power,
                          int64_t ncol, int nvec) {
    return elph_dft_accel(h, outS, inS, diagS, power, (int)ncol, h->d_nu, nvec);
}

// nch chains: b, phi laid out [sign][chain][ndim], x [chain][ndim]
int elph_launch_lambda_rhs(elph_handle_s *h, double *bS, const double *phiS, const double *xS, double dtau, int nch) {
    hipLaunchKernelGGL(k_lambda_rhs, dim3((unsigned)h->L, 2, (unsigned)nch), dim3(WAVE), 0, h->stream, bS, phiS, xS, h->d_lam,
                       (int)h->N, (int)h->L, dtau);
    return check_launch("k_lambda_rhs");
}

int elph_launch_force_holstein(elph_handle_s *h, double *FS, const double *XS, const double *phiS, const double *xS, double dtau,
                               int nch) {
    ModelDev m = elph_model_dev(h);
    const size_t shm = 2 * (size_t)h->N * sizeof(double);
    DISPATCH_NPL(gen_npl(h), {
        hipLaunchKernelGGL((k_force_holstein<NPL>), dim3((unsigned)h->L, (unsigned)nch), dim3((unsigned)gen_bs(h)), shm, h->stream, FS,
                           XS, phiS, xS, h->d_lam, m, dtau);
    });
    return check_launch("k_force_holstein");
}

int elph_launch_dmdx_holstein(elph_handle_s *h, double *FS, const double *uS, const double *vS, const double *xS, double dtau,
                              double scale, int nch) {
    ModelDev m = elph_model_dev(h);
    const size_t shm = 2 * (size_t)h->N * sizeof(double);
    DISPATCH_NPL(gen_npl(h), {
        hipLaunchKernelGGL((k_dmdx_holstein<NPL>), dim3((unsigned)h->L, (unsigned)nch), dim3((unsigned)gen_bs(h)), shm, h->stream, FS, uS, vS, xS,
                           h->d_lam, m, dtau, scale);
    });
    return check_launch("k_dmdx_holstein");
}

int elph_launch_force_ssh(elph_handle_s *h, double *q, const double *XS, const double *US, int nch) {
    ModelDev m = elph_model_dev(h);
    const size_t shm = 4 * (size_t)h->N * sizeof(double);
    DISPATCH_NPL(gen_npl(h), {
        hipLaunchKernelGGL((k_force_ssh<NPL>), dim3((unsigned)h->L, (unsigned)nch), dim3((unsigned)gen_bs(h)), shm, h->stream, q, XS, m, US, nch);
    });
    return check_launch("k_force_ssh");
}

// pieces of elph_launch_cg_init for the step-wise API (A x0 expected in d_tmp)
int elph_launch_cg_init_only(elph_handle_s *h, int nrhs) {
    CgBufs B = make_bufs(h, nrhs);
    h->ap_count = 0;
    const size_t P = (size_t)h->cap_rhs * (size_t)h->L * (size_t)h->npl;
    double *bb = h->d_part + 3 * P;
    hipLaunchKernelGGL(k_cg_init, dim3((unsigned)h->L, (unsigned)nrhs), dim3(WAVE), 0, h->stream, B, h->d_b, h->d_tmp, bb,
                       (int)h->N, (int)h->L);
    return check_launch("k_cg_init");
}

int elph_launch_cg_init_prec_only(elph_handle_s *h, int nrhs) {
    CgBufs B = make_bufs(h, nrhs);
    hipLaunchKernelGGL(k_cg_init_prec, dim3((unsigned)h->L, (unsigned)nrhs), dim3(WAVE), 0, h->stream, B, (int)h->N, (int)h->L);
    return check_launch("k_cg_init_prec");
}

int elph_launch_cg_state0_only(elph_handle_s *h, int nrhs) {
    CgBufs B = make_bufs(h, nrhs);
    const size_t P = (size_t)h->cap_rhs * (size_t)h->L * (size_t)h->npl;
    hipLaunchKernelGGL(k_cg_state0, dim3((unsigned)nrhs), dim3(WAVE), 0, h->stream, B, h->d_part + 3 * P, (int)h->L);
    return check_launch("k_cg_state0");
}
