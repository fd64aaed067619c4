// cg_sq16.hip — z = MᵀM p of the p/x-fused preconditioned batch iteration on the 16 x 16 square lattice (BASELINE config C: every
// production deck's path) and on the honeycomb lattice of 12 x 12 cells (BASELINE config D) with the checkerboard in REGISTERS:
// k_cg_ap_sq16_px, k_cg_ap_hc12_px (one body, a lattice policy each).
//
// What it replaces: lp4::k_cg_ap_chunk_px<4, T> (cg_fast_impl.inc), the lane-program form of the same step — per time slice and wave
// ~80 ds_read/ds_write_b64 through two LDS slabs (eight colour stages of gathered pairs) for 3.2 slices of HBM traffic: 61 us at 288
// right-hand sides where the bytes (245 MB) would take 40 us (profiles/r05/bench_precond_kernel_stats.csv).  Here lane l holds the 2 x 2
// patch of sites of the resident kernel's DPP form (sq_patch_site, cg_fast_common.h): x-even / y-even bonds pair registers of the lane,
// x-odd bonds are four DPP row rotations, y-odd bonds two DPP quad swaps + one ds_bpermute pair — no LDS allocation, no slab, ~30 vector
// instructions per checkerboard apply (uniform hopping: a colour is c (I + th P), th = sinh / cosh, c^4 folded into the consumer).  A
// slice's four values are two 16-byte loads per lane, rows of 128 bytes contiguous across a wave's lanes.
//
// Same recurrences as cg_ap_chunk_body<.., PX = true> (operator: HolsteinModels.jl:569-684 through Checkerboard.jl:57-83,149-175; the
// scalar state machine is IterativeSolvers.jl:153-234's as every k_cg_ap keeps it): one wave owns T consecutive slices of one
// right-hand side,
//     w(t) = p(t) - sg(t) CB [E(t) .* p(t-1)],      z(t) = w(t) - sg(t+1) E(t+1) .* CB^T w(t+1),      partial p.z per chunk.
// Against the lane-program kernel the results differ by rounding only (uniform hopping: one fma per site and colour instead of mul + fma).
#include "cg_fast_common.h"
#include "kpm_sq_dev.h"

namespace sq16 {

template <bool UNI>
struct Hop {
    double th, k4;                                   // UNI: sinh / cosh and cosh^4 (k4 = 1 otherwise)
    double c[UNI ? 1 : 4][UNI ? 1 : 4], s[UNI ? 1 : 4][UNI ? 1 : 4];   // per colour, per own site: (cosh, sinh) of the bond that covers it
    int yx;                                          // partner lane of the crossing half of the y-odd colour (sq_patch_ycross)
    __device__ __forceinline__ double upd(int col, int k, double v, double t) const {
        if constexpr (UNI) return v + th * t;
        else return __builtin_fma(s[UNI ? 0 : col][UNI ? 0 : k], t, c[UNI ? 0 : col][UNI ? 0 : k] * v);
    }
};

// one colour of the checkerboard on the four values of a lane's patch (layout and moves: cg_wg_dev.h, "The 16 x 16 square lattice …")
template <int COL, bool UNI>
__device__ __forceinline__ void colour(double (&v)[4], const Hop<UNI> &X) {
    if constexpr (COL == 0) {
        const double n0 = X.upd(0, 0, v[0], v[1]), n1 = X.upd(0, 1, v[1], v[0]), n2 = X.upd(0, 2, v[2], v[3]), n3 = X.upd(0, 3, v[3], v[2]);
        v[0] = n0; v[1] = n1; v[2] = n2; v[3] = n3;
    } else if constexpr (COL == 1) {
        const double t1 = dpp_f64<0x12E>(v[0]), t3 = dpp_f64<0x12E>(v[2]);     // row_ror:14 = lane + 2
        const double t0 = dpp_f64<0x122>(v[1]), t2 = dpp_f64<0x122>(v[3]);     // row_ror:2  = lane - 2
        v[0] = X.upd(1, 0, v[0], t0); v[1] = X.upd(1, 1, v[1], t1); v[2] = X.upd(1, 2, v[2], t2); v[3] = X.upd(1, 3, v[3], t3);
    } else if constexpr (COL == 2) {
        const double n0 = X.upd(2, 0, v[0], v[2]), n2 = X.upd(2, 2, v[2], v[0]), n1 = X.upd(2, 1, v[1], v[3]), n3 = X.upd(2, 3, v[3], v[1]);
        v[0] = n0; v[1] = n1; v[2] = n2; v[3] = n3;
    } else {
        const double c0 = __shfl(v[0], X.yx, WAVE), c1 = __shfl(v[1], X.yx, WAVE);
        const double t2 = dpp_f64<0xB1>(v[2]), t3 = dpp_f64<0xB1>(v[3]);       // quad_perm [1,0,3,2]
        v[2] = X.upd(3, 2, v[2], t2); v[3] = X.upd(3, 3, v[3], t3);
        v[0] = X.upd(3, 0, v[0], c0); v[1] = X.upd(3, 1, v[1], c1);
    }
}

// forward sweep of a (if doA) and reverse sweep of b in the same four stages: two independent dependency chains for the scheduler
template <bool UNI>
__device__ __forceinline__ void sweep_fr(double (&a)[4], double (&b)[4], const Hop<UNI> &X, bool doA) {
    if (doA) colour<0, UNI>(a, X);
    colour<3, UNI>(b, X);
    if (doA) colour<1, UNI>(a, X);
    colour<2, UNI>(b, X);
    if (doA) colour<2, UNI>(a, X);
    colour<1, UNI>(b, X);
    if (doA) colour<3, UNI>(a, X);
    colour<0, UNI>(b, X);
}
template <bool UNI>
__device__ __forceinline__ void sweep_ff(double (&a)[4], double (&b)[4], const Hop<UNI> &X) {
    colour<0, UNI>(a, X); colour<0, UNI>(b, X);
    colour<1, UNI>(a, X); colour<1, UNI>(b, X);
    colour<2, UNI>(a, X); colour<2, UNI>(b, X);
    colour<3, UNI>(a, X); colour<3, UNI>(b, X);
}

// ---- the lattice behind the kernel: NS values per lane, how a slice is loaded / stored, the two sweeps -------------------------------------------
// Square 16 x 16 (config C): the 2 x 2 patch of cg_wg_dev.h; registers 0, 1 are x-neighbours of one lattice row, 2, 3 of the other -> two
// 16-byte accesses per slice.
template <bool UNI>
struct LatSq16 {
    static constexpr int NS = 4, N = 256;
    using Ctx = Hop<UNI>;
    int s01, s23;
    __device__ __forceinline__ void init(int lane) { s01 = sq_patch_site(lane, 0); s23 = sq_patch_site(lane, 2); }
    static __device__ __forceinline__ bool act(int) { return true; }
    static __device__ __forceinline__ Ctx make(int lane, const ModelDev &m) {
        Ctx X;
        X.yx = sq_patch_ycross(lane);
        if constexpr (UNI) {
            X.th = m.s_uni / m.c_uni; X.k4 = (m.c_uni * m.c_uni) * (m.c_uni * m.c_uni);
        } else {
            X.th = 0.0; X.k4 = 1.0;
#pragma unroll
            for (int col = 0; col < 4; ++col)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int bd = m.sq_bond[col * N + sq_patch_site(lane, k)];
                    X.c[UNI ? 0 : col][UNI ? 0 : k] = m.c[bd];
                    X.s[UNI ? 0 : col][UNI ? 0 : k] = m.s[bd];
                }
        }
        return X;
    }
    static __device__ __forceinline__ double scale(const Ctx &X) { return X.k4; }
    __device__ __forceinline__ void ld(const double *row, double (&v)[NS]) const {
        const double2 a = *reinterpret_cast<const double2 *>(row + s01), b = *reinterpret_cast<const double2 *>(row + s23);
        v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
    }
    __device__ __forceinline__ void st(double *row, const double (&v)[NS]) const {
        *reinterpret_cast<double2 *>(row + s01) = make_double2(v[0], v[1]);
        *reinterpret_cast<double2 *>(row + s23) = make_double2(v[2], v[3]);
    }
    static __device__ __forceinline__ void fr(double (&a)[NS], double (&b)[NS], const Ctx &X, bool doA) { sweep_fr<UNI>(a, b, X, doA); }
    static __device__ __forceinline__ void ff(double (&a)[NS], double (&b)[NS], const Ctx &X) { sweep_ff<UNI>(a, b, X); }
};

// Honeycomb 12 x 12 cells (config D: 288 sites), uniform hopping: the QUAD layout of the Chebyshev recursion (kpm_sq_dev.h: hc12q_cb_apply) — lane
// 4 y + i (48 of 64) holds the cells x = 3 i + b of lattice row y, i.e. the SIX CONSECUTIVE sites 6 lane ... 6 lane + 5: three 16-byte accesses
// per slice; A-B pairs registers of the lane, B(x,y)-A(x+1,y) two DPP quad rotations, B(x,y)-A(x,y+1) six ds_bpermute pairs in one round trip;
// a colour is c (I + th P), c^3 folded into the consumer.  Lanes 48 ... 63 shadow lanes 0 ... 15 (valid addresses; they store nothing and enter no sum).
struct LatHc12 {
    static constexpr int NS = 6, N = 288;
    struct Ctx { kpmsq::HcLane T; double k3; };
    int s0;
    __device__ __forceinline__ void init(int lane) { s0 = 6 * ((lane < 48) ? lane : lane - 48); }
    static __device__ __forceinline__ bool act(int lane) { return lane < 48; }
    static __device__ __forceinline__ Ctx make(int lane, const ModelDev &m) {
        Ctx X;
        X.T.th = m.s_uni / m.c_uni;
        X.T.up = (lane < 48) ? (lane + 4) % 48 : lane;
        X.T.dn = (lane < 48) ? (lane + 44) % 48 : lane;
        X.k3 = m.c_uni * m.c_uni * m.c_uni;
        return X;
    }
    static __device__ __forceinline__ double scale(const Ctx &X) { return X.k3; }
    __device__ __forceinline__ void ld(const double *row, double (&v)[NS]) const {
        const double2 *q = reinterpret_cast<const double2 *>(row + s0);
        const double2 a = q[0], b = q[1], c = q[2];
        v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y; v[4] = c.x; v[5] = c.y;
    }
    __device__ __forceinline__ void st(double *row, const double (&v)[NS]) const {
        double2 *q = reinterpret_cast<double2 *>(row + s0);
        q[0] = make_double2(v[0], v[1]); q[1] = make_double2(v[2], v[3]); q[2] = make_double2(v[4], v[5]);
    }
    static __device__ __forceinline__ void fr(double (&a)[NS], double (&b)[NS], const Ctx &X, bool doA) {
        if (doA) kpmsq::hc12q_cb_apply<false>(a, X.T, []() {});
        kpmsq::hc12q_cb_apply<true>(b, X.T, []() {});
    }
    static __device__ __forceinline__ void ff(double (&a)[NS], double (&b)[NS], const Ctx &X) {
        kpmsq::hc12q_cb_apply<false>(a, X.T, []() {});
        kpmsq::hc12q_cb_apply<false>(b, X.T, []() {});
    }
};

template <int T, class LAT, int PF>
__device__ __forceinline__ void ap_body(const CgBufs &B, const ModelDev &m, int parity_order) {
    constexpr int N = LAT::N, NS = LAT::NS;
    const int lane = threadIdx.x, L = m.L, nch = L / T, parity = parity_order & 1;
    int rhs, ch;
    chunk_block_map((int)blockIdx.x, B.nrhs, nch, m.nchains > 0 ? m.nchains : 1, parity_order >> 1, rhs, ch);
    const int t0 = ch * T;
    const size_t ndim = (size_t)N * L;
    auto wrap = [L](int t) { return (t < 0) ? t + L : ((t >= L) ? t - L : t); };

    CgState *st2 = B.state + 2 * rhs;
    const CgState S = ld_state(st2 + parity);
    const CgParams P = B.params;
    const double *p = B.p + (size_t)rhs * ndim;                  // the ready search direction (slot 0: dft_mfma.hip, PxFuse)
    double *z = B.z + (size_t)rhs * ndim;
    const double *E = m.E + (size_t)(rhs % m.nchains) * m.E_chain_stride;

    LAT lat;
    lat.init(lane);
    const bool act = LAT::act(lane);
    auto ld_p = [&](int t, double (&v)[NS]) { lat.ld(p + (size_t)t * N, v); };
    auto ld_e = [&](int t, double (&v)[NS]) { lat.ld(E + (size_t)t * m.E_tau_stride, v); };

    // ---- every load of the prologue, independent of everything ---------------------------------------------------------------
    double Pm[NS], P0[NS], P1[NS], E0[NS], E1[NS];
    ld_p(wrap(t0 - 1), Pm); ld_p(t0, P0); ld_p(wrap(t0 + 1), P1);
    ld_e(t0, E0); ld_e(wrap(t0 + 1), E1);
    double Pr[PF][NS], Er[PF][NS];
#pragma unroll
    for (int k = 0; k < PF; ++k)
        if (k + 2 <= T) { ld_p(wrap(t0 + 2 + k), Pr[k]); ld_e(wrap(t0 + 2 + k), Er[k]); }
    const typename LAT::Ctx X = LAT::make(lane, m);
    const double kscale = LAT::scale(X);
    const double rr = reduce_partials2(B.rr + (size_t)rhs * L, L);
    const double rz = P.use_prec ? reduce_partials2(B.rz + (size_t)rhs * B.nrz, B.nrz) : rr;

    // ---- scalar control: identical in every wave of this right-hand side (the code of cg_ap_chunk_body, PX) ---------------------
    CgState *Sout = st2 + (parity ^ 1);
    if (S.done) {
        if (ch == 0 && lane == 0) *Sout = S;
        return;
    }
    const long long seq = S.seq;
    double rho = S.rho, kmin = S.kmin, eps = S.eps;
    if (seq != 0) {
        const int done = cg_stop_test(P, S, rr, seq, eps, kmin);
        if (ch == 0 && lane == 0 && P.record_hist) B.hist[(size_t)rhs * P.hist_stride + seq] = eps;
        if (done) {                                      // (x += alpha p of the last iteration: the inverse transform has applied it)
            if (ch == 0 && lane == 0) {
                CgState o = S;
                o.kmin = kmin; o.eps = eps; o.seq = seq + 1; o.iters = seq; o.done = done;
                *Sout = o;
            }
            return;
        }
        rho = rz;
    }
    auto sgk = [kscale](int t) { return (t == 0) ? -kscale : kscale; };      // sign of the slice x the factor the sweeps left out

    // (lanes that hold no sites — the honeycomb layout's lanes 48 ... 63 — sit the sweeps out altogether: no active lane reads from them, and one
    //  masked region keeps the compiler from sinking every slice's conditional store to the end of the kernel)
    double acc = 0.0;
    if (act) {
    // ---- w(t0), w(t0+1): two forward sweeps side by side ---------------------------------------------------------------------------
    double pprev[NS], pcur[NS], wprev[NS], wcur[NS], Ecur[NS];
    {
        double a[NS], b[NS];
#pragma unroll
        for (int q = 0; q < NS; ++q) { a[q] = E0[q] * Pm[q]; b[q] = E1[q] * P0[q]; }
        LAT::ff(a, b, X);
        const double ga = sgk(t0), gb = sgk(wrap(t0 + 1));
#pragma unroll
        for (int q = 0; q < NS; ++q) {
            pprev[q] = P0[q]; pcur[q] = P1[q]; Ecur[q] = E1[q];
            wprev[q] = P0[q] - ga * a[q];
            wcur[q] = P1[q] - gb * b[q];
        }
    }

    // ---- stage j: reverse sweep of w(t0+j)  ||  forward sweep of E(t0+j+1) .* p(t0+j); slices t0+2 … t0+T stream through a ring ------
#pragma unroll
    for (int j = 1; j <= T; ++j) {
        const int tj = wrap(t0 + j), tn = wrap(t0 + j + 1);
        const bool more = (j < T);
        double (&Pn)[NS] = Pr[(j - 1) % PF];
        double (&En)[NS] = Er[(j - 1) % PF];
        double a[NS], b[NS];
#pragma unroll
        for (int q = 0; q < NS; ++q) { b[q] = wcur[q]; a[q] = more ? En[q] * pcur[q] : 0.0; }
        LAT::fr(a, b, X, more);
        const double gj = sgk(tj), gn = sgk(tn);
        double zz[NS];
#pragma unroll
        for (int q = 0; q < NS; ++q) {
            zz[q] = wprev[q] - gj * Ecur[q] * b[q];                      // z(t0+j-1)
            acc += pprev[q] * zz[q];
        }
        lat.st(z + (size_t)wrap(t0 + j - 1) * N, zz);
        if (more) {
#pragma unroll
            for (int q = 0; q < NS; ++q) {
                const double pn = Pn[q];
                wprev[q] = wcur[q]; wcur[q] = pn - gn * a[q];
                pprev[q] = pcur[q]; pcur[q] = pn;
                Ecur[q] = En[q];
            }
            if (j + 1 + PF <= T) { ld_p(wrap(t0 + j + 1 + PF), Pn); ld_e(wrap(t0 + j + 1 + PF), En); }
        }
    }
    }      // if (act)
    acc = wave_sum2(acc);
    if (lane == 0) {
        B.pap[(size_t)rhs * B.npap + ch] = acc;
        if (ch == 0) {
            CgState o = S;
            o.rho = rho; o.kmin = kmin; o.eps = eps; o.seq = seq + 1; o.iters = seq; o.done = 0;
            *Sout = o;
        }
    }
}

// WPE waves per SIMD: 4 (128 registers) is what a ring of two slices needs with uniform hopping; disordered hopping keeps 32 (cosh, sinh)
// pairs per lane and runs at 2
template <int T, bool UNI, int PF, int WPE>
__global__ void __launch_bounds__(WAVE) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) k_cg_ap_sq16_px(CgBufs B, ModelDev m, int parity_order) {
    ap_body<T, LatSq16<UNI>, PF>(B, m, parity_order);
}
// the honeycomb lattice of 12 x 12 cells (config D): six values per lane — a ring of PF slices at WPE waves per SIMD
template <int T, int PF, int WPE>
__global__ void __launch_bounds__(WAVE) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) k_cg_ap_hc12_px(CgBufs B, ModelDev m, int parity_order) {
    ap_body<T, LatHc12, PF>(B, m, parity_order);
}

}  // namespace sq16

// Does the p/x-fused k_cg_ap of this handle run in the register-exchange form?  Holstein on the 16 x 16 square lattice in the reference's
// colouring (detect_square: sq_P = 2), a template chunk length.  ELPH_SQ16_AP=0: the lane-program kernel (A/B; read per call).
bool elph_sq16_ap_usable(const elph_handle_s *h, int T) {
    const char *e = getenv("ELPH_SQ16_AP");
    if (e && e[0] == '0') return false;
    if (h->kind != ELPH_MODEL_HOLSTEIN) return false;
    const bool sq = h->sq_P == 2 && h->N == 256 && h->d_sq_bond, hc = h->hc12 && h->hc_uniform && h->N == 288;      // (config C; config D, uniform hopping)
    if (!sq && !hc) return false;
    return h->L % T == 0 && (T == 20 || T == 16 || T == 10 || T == 8 || T == 5 || T == 4 || T == 2);
}

int elph_sq16_cg_ap_px(elph_handle_s *h, const CgBufs &B, int nrhs, int parity) {
    ModelDev m = elph_model_dev(h);
    const int T = (int)(h->L / B.npap);
    if (!elph_sq16_ap_usable(h, T) || (int)h->L != T * B.npap) { elph_set_error("k_cg_ap_sq16_px: not planned for this handle"); return ELPH_E_STATE; }
    static const int xcd_order = []() { const char *e = getenv("ELPH_CHUNK_ORDER"); return (e && e[0] == '0') ? 0 : 1; }();
    const int po = parity | ((xcd_order && h->solo_chain < 0) ? 2 : 0);
    const dim3 grid((unsigned)(nrhs * B.npap));
    // (ring depth / waves per SIMD: ELPH_SQ16_SHAPE=<depth><waves>, e.g. 24 — measurement only)
    const int shape = []() { const char *e = getenv("ELPH_SQ16_SHAPE"); return e ? atoi(e) : 0; }();      // (read per call: in-process A/B)
#define SQ16_LAUNCH(TT)                                                                                                            \
    do {                                                                                                                            \
        if (h->hc12 && shape == 32) hipLaunchKernelGGL((sq16::k_cg_ap_hc12_px<TT, 3, 2>), grid, dim3(WAVE), 0, h->stream, B, m, po);   \
        else if (h->hc12 && shape == 22) hipLaunchKernelGGL((sq16::k_cg_ap_hc12_px<TT, 2, 2>), grid, dim3(WAVE), 0, h->stream, B, m, po); \
        else if (h->hc12) hipLaunchKernelGGL((sq16::k_cg_ap_hc12_px<TT, 2, 3>), grid, dim3(WAVE), 0, h->stream, B, m, po);           \
        else if (!m.uniform) hipLaunchKernelGGL((sq16::k_cg_ap_sq16_px<TT, false, 2, 2>), grid, dim3(WAVE), 0, h->stream, B, m, po);     \
        else if (shape == 24) hipLaunchKernelGGL((sq16::k_cg_ap_sq16_px<TT, true, 2, 4>), grid, dim3(WAVE), 0, h->stream, B, m, po); \
        else if (shape == 33) hipLaunchKernelGGL((sq16::k_cg_ap_sq16_px<TT, true, 3, 3>), grid, dim3(WAVE), 0, h->stream, B, m, po); \
        else if (shape == 42) hipLaunchKernelGGL((sq16::k_cg_ap_sq16_px<TT, true, 4, 2>), grid, dim3(WAVE), 0, h->stream, B, m, po); \
        else hipLaunchKernelGGL((sq16::k_cg_ap_sq16_px<TT, true, 4, 3>), grid, dim3(WAVE), 0, h->stream, B, m, po);                 \
    } while (0)
    switch (T) {
        case 20: SQ16_LAUNCH(20); break;
        case 16: SQ16_LAUNCH(16); break;
        case 10: SQ16_LAUNCH(10); break;
        case 8: SQ16_LAUNCH(8); break;
        case 5: SQ16_LAUNCH(5); break;
        case 4: SQ16_LAUNCH(4); break;
        default: SQ16_LAUNCH(2); break;
    }
#undef SQ16_LAUNCH
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { elph_set_error("launch k_cg_ap_sq16_px failed: %s", hipGetErrorString(e)); return ELPH_E_HIP; }
    return ELPH_OK;
}
