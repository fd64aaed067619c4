// pgrid_dev.h — the PGRID layout: an even-L square lattice LARGER than 16 x 16 in the registers of ONE wavefront.
//
// The GRID layout (cg_fast_common.h) gives a lane a 2 x 2 patch: 64 lanes carry at most 16 x 16 sites.  The lattices people actually
// simulate are larger (L = 20, 24, 32: 400 ... 1024 sites per time slice), and for them the library had only the generic kernels — a
// workgroup of up to 1024 threads per slice, every checkerboard colour an LDS round trip and a barrier.  Here a lane holds a PX x PY
// patch (PX, PY even): lane (X, Y) = (l % GX, l / GX), GX = L / PX, GY = L / PY, register q = cx + PX cy is the site
// (PX X + cx) + L (PY Y + cy).  The reference's colouring of the square lattice (Checkerboard.jl:57-141 through lattice.py; recognised
// by detect_square, elph_api.hip) is [x-even | x-odd | y-even | y-odd] bonds: the even colours and the inner pairs of the odd colours
// pair registers of one lane; only the patch EDGES cross — PY values per direction for x-odd, PX for y-odd, by ds_bpermute, all issued
// before the first is used.  A 4 x 4 patch does 16 fma per colour and moves 4 + 4 values in two of the four colours: a quarter of the
// exchange per fma of the 2 x 2 patch, and no barrier anywhere.
//   L = 32, 28: 4 x 4 patches (64 / 49 lanes, 16 registers per vector)      L = 24, 18: 2 x 6 (48 / 27 lanes, 12 registers)
//   L = 20:     2 x 4 (50 lanes, 8 registers)                               L = 30: 2 x 10 (45 lanes, 20), L = 36: 4 x 6 (54 lanes, 24)
// Uniform hopping only (one (cosh, sinh) for every bond — the decks): a colour is c (I + th P) with th = sinh / cosh, the caller
// scales by c^4 once.
#pragma once
#include "elph_internal.h"

namespace pgrid {

constexpr int WAVE_ = ELPH_WAVE;

struct Ctx {
    double th, ks;           // tanh of the bond angle; the factor taken out of the colours: c^4 (square), c^3 (honeycomb)
    int xu, xd, yu, yd;      // lanes of the patches X + 1, X - 1, Y + 1, Y - 1 (cyclic); idle lanes: themselves
    int du, dd;              // lanes of the patches (X - 1, Y + 1) and (X + 1, Y - 1): the diagonal bonds of the triangular lattice
    // MULTI-WAVE slices (round 6: square lattices whose patches need more than 64 lanes — L = 22, 26, 34, 38 as 2 x 2 patches on 2 ... 6 wavefronts,
    // L = 40 ... 64 as 4 x 4 patches on 2 ... 4): the "lanes" above are thread indices of the slice's NT = 64 NW threads, and the patch edges
    // cross through LDS instead of ds_bpermute — xb: four exchange buffers of 2 PBMAX x NT doubles (x-odd / y-odd colour x forward / reverse sweep:
    // with a buffer of its own per colour and direction ONE barrier per crossing colour orders everything), nullptr for one wavefront
    double *xb;
    int nt, me;              // threads of the slice; this thread's index in it
    int bmask;               // 3: four buffers; 1: the reverse sweeps share the forward sweeps' two (a kernel whose directions never alternate without a barrier: the Chebyshev series)
    // HOPPING DISORDER (round 6; square patches, Sq<..., UNI = false>): the (cosh, sinh) of the bonds this thread touches, one entry per bond, at
    // tab[(col_base<COL> + slot) nt + me] (LDS, filled by the kernel from the bond tables through the site -> bond map of the colouring; ColDims); a
    // colour is then v_i <- c v_i + s v_j with the bond's own pair, nothing is factored out (ks = 1).  nullptr: uniform hopping.
    const double2 *tab;
};

template <int PX, int PY>
__host__ __device__ __forceinline__ int site(int lane, int q, int L) {
    const int GX = L / PX, GY = L / PY;
    const int l = (lane < GX * GY) ? lane : 0, X = l % GX, Y = l / GX;
    const int cx = q % PX, cy = q / PX;
    return (PX * X + cx) + L * (PY * Y + cy);
}

template <int PX, int PY>
__device__ __forceinline__ Ctx ctx(int lane, int L, double c, double s) {
    const int GX = L / PX, GY = L / PY;
    Ctx X;
    X.xb = nullptr; X.nt = WAVE_; X.me = lane; X.bmask = 3; X.tab = nullptr;
    X.th = s / c; X.ks = (c * c) * (c * c);
    if (lane < GX * GY) {
        const int x = lane % GX, y = lane / GX;
        X.xu = (x + 1) % GX + GX * y; X.xd = (x + GX - 1) % GX + GX * y;
        X.yu = x + GX * ((y + 1) % GY); X.yd = x + GX * ((y + GY - 1) % GY);
        X.du = (x + GX - 1) % GX + GX * ((y + 1) % GY); X.dd = (x + 1) % GX + GX * ((y + GY - 1) % GY);
    } else {
        X.xu = X.xd = X.yu = X.yd = X.du = X.dd = lane;
    }
    return X;
}

// The geometry of one colour on a PX x PY patch, and where its bonds sit in the (cosh, sinh) table of the disorder variant: ONE entry per bond a
// thread touches — a pair inside the patch is one entry, an edge site has the entry of its crossing bond (4 x 4: 8 + 12 + 8 + 12 = 40 entries
// where one per site and colour would be 64 — 40 KB per wavefront instead of 64: four blocks per CU again).
template <int PX, int PY, int COL> struct ColDims {
    static constexpr bool ALONG_X = (COL < 2), ODD = (COL & 1);
    static constexpr int PA = ALONG_X ? PX : PY, PB = ALONG_X ? PY : PX, SA = ALONG_X ? 1 : PX, SB = ALONG_X ? PX : 1;
    static constexpr int INNER = ODD ? PB * (PA / 2 - 1) : PB * (PA / 2);        // pairs inside the patch
    static constexpr int ENTRIES = ODD ? INNER + 2 * PB : INNER;
    __host__ __device__ static constexpr int pair_slot(int a, int b) { return ODD ? b * (PA / 2 - 1) + (a - 1) / 2 : b * (PA / 2) + a / 2; }
    __host__ __device__ static constexpr int edge_slot(bool hi, int b) { return INNER + (hi ? 0 : PB) + b; }
};
template <int PX, int PY, int COL> __host__ __device__ constexpr int col_base() {
    if constexpr (COL == 0) return 0;
    else return col_base<PX, PY, COL - 1>() + ColDims<PX, PY, COL - 1>::ENTRIES;
}
template <int PX, int PY> __host__ __device__ constexpr int tab_entries() { return col_base<PX, PY, 3>() + ColDims<PX, PY, 3>::ENTRIES; }

// one colour of the checkerboard on the PX x PY values of a lane: v <- (I + th P_colour) v   (UNI = false: v_i <- c v_i + s v_j per bond, Ctx::tab)
template <int PX, int PY, int COL, int BUF = 0, bool UNI = true>
__device__ __forceinline__ void colour(double (&v)[PX * PY], const Ctx &X) {
    constexpr bool ALONG_X = (COL < 2), ODD = (COL & 1);
    constexpr int PA = ALONG_X ? PX : PY;        // patch extent along the colour's direction
    constexpr int PB = ALONG_X ? PY : PX;        // ... and across it
    constexpr int SA = ALONG_X ? 1 : PX;         // register strides along / across
    constexpr int SB = ALONG_X ? PX : 1;
    // (cosh, sinh) of the bond in table slot `slot` of this colour
    using D = ColDims<PX, PY, COL>;
    auto cs = [&X](int slot) { return UNI ? make_double2(1.0, X.th) : X.tab[(size_t)(col_base<PX, PY, COL>() + slot) * X.nt + X.me]; };
    if constexpr (ODD) {
        // the edge pairs cross to the neighbouring patches: my last column pairs with the first column of the patch above, my first
        // column with the last column of the patch below
        const int up = ALONG_X ? X.xu : X.yu, dn = ALONG_X ? X.xd : X.yd;
        double fu[PB], fd[PB];
        if (X.xb) {
            // several wavefronts per slice: the edges through LDS (buffer BUF of this colour and sweep direction), one barrier
            constexpr int PBM = (PX > PY) ? PX : PY;
            double *xb = X.xb + (size_t)(BUF & X.bmask) * 2 * PBM * X.nt;
#pragma unroll
            for (int b = 0; b < PB; ++b) {
                xb[(2 * b) * X.nt + X.me] = v[0 * SA + b * SB];
                xb[(2 * b + 1) * X.nt + X.me] = v[(PA - 1) * SA + b * SB];
            }
            __syncthreads();
#pragma unroll
            for (int b = 0; b < PB; ++b) {
                fu[b] = xb[(2 * b) * X.nt + up];
                fd[b] = xb[(2 * b + 1) * X.nt + dn];
            }
        } else {
#pragma unroll
            for (int b = 0; b < PB; ++b) {
                fu[b] = __shfl(v[0 * SA + b * SB], up, WAVE_);
                fd[b] = __shfl(v[(PA - 1) * SA + b * SB], dn, WAVE_);
            }
        }
        // the inner pairs (1,2), (3,4), ... while the crossings fly
#pragma unroll
        for (int b = 0; b < PB; ++b)
#pragma unroll
            for (int a = 1; a + 1 < PA; a += 2) {
                const int i = a * SA + b * SB, j = i + SA;
                if constexpr (UNI) {
                    const double ni = v[i] + X.th * v[j], nj = v[j] + X.th * v[i];
                    v[i] = ni; v[j] = nj;
                } else {
                    const double2 t = cs(D::pair_slot(a, b));
                    const double ni = t.x * v[i] + t.y * v[j], nj = t.x * v[j] + t.y * v[i];
                    v[i] = ni; v[j] = nj;
                }
            }
#pragma unroll
        for (int b = 0; b < PB; ++b) {
            if constexpr (UNI) {
                v[(PA - 1) * SA + b * SB] += X.th * fu[b];
                v[0 * SA + b * SB] += X.th * fd[b];
            } else {
                const double2 tu = cs(D::edge_slot(true, b)), td = cs(D::edge_slot(false, b));
                v[(PA - 1) * SA + b * SB] = tu.x * v[(PA - 1) * SA + b * SB] + tu.y * fu[b];
                v[0 * SA + b * SB] = td.x * v[0 * SA + b * SB] + td.y * fd[b];
            }
        }
    } else {
#pragma unroll
        for (int b = 0; b < PB; ++b)
#pragma unroll
            for (int a = 0; a + 1 < PA; a += 2) {
                const int i = a * SA + b * SB, j = i + SA;
                if constexpr (UNI) {
                    const double ni = v[i] + X.th * v[j], nj = v[j] + X.th * v[i];
                    v[i] = ni; v[j] = nj;
                } else {
                    const double2 t = cs(D::pair_slot(a, b));
                    const double ni = t.x * v[i] + t.y * v[j], nj = t.x * v[j] + t.y * v[i];
                    v[i] = ni; v[j] = nj;
                }
            }
    }
}

// the whole checkerboard (REVERSE: its transpose — the colours in reverse order; every colour is symmetric), WITHOUT the factor c^4
template <int PX, int PY, bool REVERSE, bool UNI = true>
__device__ __forceinline__ void sweep(double (&v)[PX * PY], const Ctx &X) {
    if constexpr (!REVERSE) { colour<PX, PY, 0, 0, UNI>(v, X); colour<PX, PY, 1, 0, UNI>(v, X); colour<PX, PY, 2, 0, UNI>(v, X); colour<PX, PY, 3, 1, UNI>(v, X); }
    else                    { colour<PX, PY, 3, 3, UNI>(v, X); colour<PX, PY, 2, 0, UNI>(v, X); colour<PX, PY, 1, 2, UNI>(v, X); colour<PX, PY, 0, 0, UNI>(v, X); }
}

// ---- honeycomb: L x L two-site cells (site = 2 (x + L y) + orbital) in the reference's colouring [A-B of a cell | B(x,y)-A(x+1,y) |
// B(x,y)-A(x,y+1)] (detect_honeycomb, elph_api.hip), PX x PY CELLS per lane: register q = 2 (cx + PX cy) + orbital.  A-B pairs registers
// of the lane; the other two colours pair registers of the lane inside the patch and cross at its edge — PY resp. PX values each way.
// No even/odd structure: any patch shape that divides L.  (Lattices of up to 16 x 16 cells have the HGRID form, cg_fast_common.h.)
template <int PX, int PY>
__host__ __device__ __forceinline__ int hsite(int lane, int q, int L) {
    const int GX = L / PX, GY = L / PY;
    const int l = (lane < GX * GY) ? lane : 0, X = l % GX, Y = l / GX;
    const int c = q >> 1, cx = c % PX, cy = c / PX;
    return 2 * ((PX * X + cx) + L * (PY * Y + cy)) + (q & 1);
}
template <int PX, int PY, int COL, int BUF = 0>
__device__ __forceinline__ void hcolour(double (&v)[2 * PX * PY], const Ctx &X) {
    if constexpr (COL == 0) {
#pragma unroll
        for (int c = 0; c < PX * PY; ++c) {
            const double na = v[2 * c] + X.th * v[2 * c + 1], nb = v[2 * c + 1] + X.th * v[2 * c];
            v[2 * c] = na; v[2 * c + 1] = nb;
        }
    } else {
        constexpr bool ALONG_X = (COL == 1);
        constexpr int PA = ALONG_X ? PX : PY, PB = ALONG_X ? PY : PX;
        constexpr int SA = ALONG_X ? 1 : PX, SB = ALONG_X ? PX : 1;          // cell strides along / across
        const int up = ALONG_X ? X.xu : X.yu, dn = ALONG_X ? X.xd : X.yd;
        double fu[PB], fd[PB];
        if (X.xb) {      // several wavefronts per slice: the edges through LDS (the square lattice's scheme, `colour` above)
            constexpr int PBM = (PX > PY) ? PX : PY;
            double *xb = X.xb + (size_t)(BUF & X.bmask) * 2 * PBM * X.nt;
#pragma unroll
            for (int b = 0; b < PB; ++b) {
                xb[(2 * b) * X.nt + X.me] = v[2 * (0 * SA + b * SB)];
                xb[(2 * b + 1) * X.nt + X.me] = v[2 * ((PA - 1) * SA + b * SB) + 1];
            }
            __syncthreads();
#pragma unroll
            for (int b = 0; b < PB; ++b) {
                fu[b] = xb[(2 * b) * X.nt + up];
                fd[b] = xb[(2 * b + 1) * X.nt + dn];
            }
        } else {
#pragma unroll
            for (int b = 0; b < PB; ++b) {
                fu[b] = __shfl(v[2 * (0 * SA + b * SB)], up, WAVE_);                    // the A site of the first cell of the patch above
                fd[b] = __shfl(v[2 * ((PA - 1) * SA + b * SB) + 1], dn, WAVE_);         // the B site of the last cell of the patch below
            }
        }
#pragma unroll
        for (int b = 0; b < PB; ++b)
#pragma unroll
            for (int a = 0; a + 1 < PA; ++a) {                                      // B(a) - A(a + 1) inside the patch
                const int i = 2 * (a * SA + b * SB) + 1, j = 2 * ((a + 1) * SA + b * SB);
                const double ni = v[i] + X.th * v[j], nj = v[j] + X.th * v[i];
                v[i] = ni; v[j] = nj;
            }
#pragma unroll
        for (int b = 0; b < PB; ++b) {
            v[2 * ((PA - 1) * SA + b * SB) + 1] += X.th * fu[b];
            v[2 * (0 * SA + b * SB)] += X.th * fd[b];
        }
    }
}

// ---- triangular: the square lattice's sites with a third bond direction (1, -1) — (x, y) - (x - 1, y + 1) — in the reference's colouring
// [x-even | x-odd | y-even | diagonal from even y | y-odd | diagonal from odd y] (the order Checkerboard.jl's colouring gives the bond
// definitions of examples/holstein_hmc_triangular.toml; recognised by detect_triangular, elph_api.hip).  PX x PY patches as for the
// square lattice; a diagonal from an even row stays inside the patch rows (cy, cy + 1) and crosses only in x (the column cx = 0 to the
// patch X - 1); a diagonal from the patch's LAST row goes to the patch row above: PX - 1 values from (X, Y + 1), one — the corner — from
// (X - 1, Y + 1).
template <int PX, int PY, int PAR>      // PAR = 0: diagonals from even rows, 1: from odd rows
__device__ __forceinline__ void tdiag(double (&v)[PX * PY], const Ctx &X) {
    // rows (cy, cy + 1) with cy = PAR, PAR + 2, ... inside the patch: (cx, cy) - (cx - 1, cy + 1)
    // crossings in x for the inner row pairs: my (0, cy) with (PX - 1, cy + 1) of the patch X - 1; my (PX - 1, cy + 1) with (0, cy) of X + 1
    constexpr int NR = (PAR == 0) ? PY / 2 : PY / 2 - 1;                                     // inner row pairs: cy = PAR + 2 k, cy + 1 < PY
    double fxd[NR > 0 ? NR : 1], fxu[NR > 0 ? NR : 1];
#pragma unroll
    for (int k = 0; k < NR; ++k) {
        const int cy = PAR + 2 * k;
        fxd[k] = __shfl(v[(PX - 1) + PX * (cy + 1)], X.xd, WAVE_);       // the patch X - 1: its (PX - 1, cy + 1)
        fxu[k] = __shfl(v[0 + PX * cy], X.xu, WAVE_);                    // the patch X + 1: its (0, cy)
    }
    // the last row (PAR = 1 only: cy = PY - 1 is odd) pairs with row 0 of the patch row above
    double fyu[PX], fyd[PX];
    if constexpr (PAR == 1) {
#pragma unroll
        for (int cx = 1; cx < PX; ++cx) fyu[cx] = __shfl(v[(cx - 1) + PX * 0], X.yu, WAVE_);          // (cx, PY-1) - (cx-1, 0) of (X, Y+1)
        fyu[0] = __shfl(v[(PX - 1) + PX * 0], X.du, WAVE_);                                           // (0, PY-1) - (PX-1, 0) of (X-1, Y+1)
#pragma unroll
        for (int cx = 0; cx + 1 < PX; ++cx) fyd[cx] = __shfl(v[(cx + 1) + PX * (PY - 1)], X.yd, WAVE_);   // (cx, 0) - (cx+1, PY-1) of (X, Y-1)
        fyd[PX - 1] = __shfl(v[0 + PX * (PY - 1)], X.dd, WAVE_);                                      // (PX-1, 0) - (0, PY-1) of (X+1, Y-1)
    }
    // inside the lane
#pragma unroll
    for (int k = 0; k < NR; ++k) {
        const int cy = PAR + 2 * k;
#pragma unroll
        for (int cx = 1; cx < PX; ++cx) {
            const int i = cx + PX * cy, j = (cx - 1) + PX * (cy + 1);
            const double ni = v[i] + X.th * v[j], nj = v[j] + X.th * v[i];
            v[i] = ni; v[j] = nj;
        }
    }
#pragma unroll
    for (int k = 0; k < NR; ++k) {
        const int cy = PAR + 2 * k;
        v[0 + PX * cy] += X.th * fxd[k];
        v[(PX - 1) + PX * (cy + 1)] += X.th * fxu[k];
    }
    if constexpr (PAR == 1) {
#pragma unroll
        for (int cx = 0; cx < PX; ++cx) {
            v[cx + PX * (PY - 1)] += X.th * fyu[cx];
            v[cx + PX * 0] += X.th * fyd[cx];
        }
    }
}

// ---- the two lattices behind one interface: NS registers per vector, site(), ctx(), sweep<REVERSE>() -----------------------------------
// NW: wavefronts per time slice (1: the whole slice in one wavefront, edges by ds_bpermute; > 1: NT = 64 NW threads, edges through LDS)
// UNI_ = false: hopping disorder — a (cosh, sinh) pair per bond from a table in LDS (Ctx::tab; TAB_BYTES of dynamic LDS per block)
template <int PX_, int PY_, int NW_ = 1, bool UNI_ = true> struct Sq {
    static constexpr int PX = PX_, PY = PY_, NS = PX_ * PY_, NW = NW_;
    static constexpr bool DIS = !UNI_;
    static constexpr int NCOL = 4;
    static constexpr size_t TAB_BYTES = UNI_ ? 0 : (size_t)tab_entries<PX_, PY_>() * NW_ * WAVE_ * sizeof(double2);
    static constexpr int XB_DOUBLES = (NW_ > 1) ? 4 * 2 * ((PX_ > PY_) ? PX_ : PY_) * NW_ * WAVE_ : 1;      // the four exchange buffers of one slice
    __host__ __device__ static int lanes(int L) { return (L / PX) * (L / PY); }
    __host__ __device__ static int site_of(int lane, int q, int L) { return site<PX, PY>(lane, q, L); }
    static constexpr int XB2_DOUBLES = (NW_ > 1) ? 2 * 2 * ((PX_ > PY_) ? PX_ : PY_) * NW_ * WAVE_ : 1;     // ... two of them (bmask = 1)
    __device__ static Ctx make_ctx(int lane, int L, double c, double s, double *xb = nullptr, int bmask = 3) {
        Ctx X = ctx<PX, PY>(lane, L, c, s);
        if (NW > 1) { X.xb = xb; X.nt = NW * WAVE_; X.bmask = bmask; }
        if (DIS) { X.ks = 1.0; X.th = 0.0; }
        return X;
    }
    template <bool REVERSE> __device__ static void apply(double (&v)[NS], const Ctx &X) { sweep<PX, PY, REVERSE, UNI_>(v, X); }
};
template <int PX_, int PY_, int NW_ = 1> struct Hc {
    static constexpr int PX = PX_, PY = PY_, NS = 2 * PX_ * PY_, NW = NW_;
    static constexpr bool DIS = false;
    static constexpr int NCOL = 3;
    static constexpr size_t TAB_BYTES = 0;
    static constexpr int XB_DOUBLES = (NW_ > 1) ? 4 * 2 * ((PX_ > PY_) ? PX_ : PY_) * NW_ * WAVE_ : 1, XB2_DOUBLES = (NW_ > 1) ? XB_DOUBLES / 2 : 1;
    __host__ __device__ static int lanes(int L) { return (L / PX) * (L / PY); }
    __host__ __device__ static int site_of(int lane, int q, int L) { return hsite<PX, PY>(lane, q, L); }
    __device__ static Ctx make_ctx(int lane, int L, double c, double s, double *xb = nullptr, int bmask = 3) {
        Ctx X = ctx<PX, PY>(lane, L, c, s);
        X.ks = c * c * c;                      // three colours
        if (NW > 1) { X.xb = xb; X.nt = NW * WAVE_; X.bmask = bmask; }
        return X;
    }
    template <bool REVERSE> __device__ static void apply(double (&v)[NS], const Ctx &X) {
        if constexpr (!REVERSE) { hcolour<PX, PY, 0>(v, X); hcolour<PX, PY, 1, 0>(v, X); hcolour<PX, PY, 2, 1>(v, X); }
        else                    { hcolour<PX, PY, 2, 3>(v, X); hcolour<PX, PY, 1, 2>(v, X); hcolour<PX, PY, 0>(v, X); }
    }
};

inline bool pick_patch(int L, int *PX, int *PY);
template <int PX_, int PY_> struct Tri {
    static constexpr int PX = PX_, PY = PY_, NS = PX_ * PY_, NW = 1, XB_DOUBLES = 1, XB2_DOUBLES = 1;
    static constexpr bool DIS = false;
    static constexpr int NCOL = 6;
    static constexpr size_t TAB_BYTES = 0;
    __host__ __device__ static int lanes(int L) { return (L / PX) * (L / PY); }
    __host__ __device__ static int site_of(int lane, int q, int L) { return site<PX, PY>(lane, q, L); }
    __device__ static Ctx make_ctx(int lane, int L, double c, double s, double * = nullptr, int = 3) {
        Ctx X = ctx<PX, PY>(lane, L, c, s);
        X.ks = (c * c * c) * (c * c * c);      // six colours
        return X;
    }
    template <bool REVERSE> __device__ static void apply(double (&v)[NS], const Ctx &X) {
        if constexpr (!REVERSE) {
            colour<PX, PY, 0>(v, X); colour<PX, PY, 1>(v, X); colour<PX, PY, 2>(v, X); tdiag<PX, PY, 0>(v, X); colour<PX, PY, 3>(v, X); tdiag<PX, PY, 1>(v, X);
        } else {
            tdiag<PX, PY, 1>(v, X); colour<PX, PY, 3>(v, X); tdiag<PX, PY, 0>(v, X); colour<PX, PY, 2>(v, X); colour<PX, PY, 1>(v, X); colour<PX, PY, 0>(v, X);
        }
    }
};
// The patch for an even-L triangular lattice (any even L from 4: no other register form exists for it).
inline bool pick_tpatch(int L, int *PX, int *PY) {
    if (L < 4 || (L & 1)) return false;
    if (L <= 16) { *PX = 2; *PY = 2; return true; }
    // (only the shapes pgrid.hip instantiates for the triangular sweep: 2 x 4, 2 x 6, 4 x 4 — a triangular 30 x 30 or 36 x 36 keeps the generic kernels)
    return pick_patch(L, PX, PY) && ((*PX == 2 && (*PY == 4 || *PY == 6)) || (*PX == 4 && *PY == 4));
}

// The multi-wave cell patch for an L x L honeycomb lattice that has no single-wave one (round 6): 3 x 3 cells per thread on 2, 3 or 4 wavefronts
// (L = 27, 30, 33, 36, 39, 42, 45, 48), else 2 x 2 cells on 2, 3 or 4 (L = 22, 26, 28, 32) — the shapes pgrid.hip instantiates.
inline bool pick_hpatch_mw(int L, int *PX, int *PY, int *NW) {
    if (L <= 16) return false;
    for (int P : {3, 2}) {
        if (L % P) continue;
        const int lanes = (L / P) * (L / P), nw = (lanes + 63) / 64;
        if (nw >= 2 && nw <= 4) { *PX = P; *PY = P; *NW = nw; return true; }
    }
    return false;
}

// The cell patch for an L x L honeycomb lattice beyond 16 x 16 cells (false: none).
inline bool pick_hpatch(int L, int *PX, int *PY) {
    switch (L) {
        case 18: *PX = 3; *PY = 2; return true;      // 6 x 9 lanes, 12 registers per vector
        case 20: *PX = 4; *PY = 2; return true;      // 5 x 10 lanes, 16
        case 21: case 24: *PX = 3; *PY = 3; return true;      // 7 x 7 / 8 x 8 lanes, 18
        default: return false;
    }
}

// The multi-wave patch for an even-L square lattice that has no single-wave one (round 6): 2 x 2 patches on 2, 3, 5 or 6 wavefronts (L = 22, 26, 34,
// 38), 4 x 4 patches on 2, 3 or 4 (L = 40, 44, 48, 52, 56, 60, 64) — the shapes pgrid.hip instantiates.  *NW = wavefronts per slice.
inline bool pick_patch_mw(int L, int *PX, int *PY, int *NW) {
    if (L < 18 || (L & 1)) return false;
    if (L % 4 == 0 && L >= 40 && L <= 64) {
        const int lanes = (L / 4) * (L / 4), nw = (lanes + 63) / 64;
        if (nw >= 2 && nw <= 4) { *PX = 4; *PY = 4; *NW = nw; return true; }
    }
    if (L == 22 || L == 26 || L == 34 || L == 38) {
        const int lanes = (L / 2) * (L / 2), nw = (lanes + 63) / 64;      // 121, 169, 289, 361 -> 2, 3, 5, 6
        *PX = 2; *PY = 2; *NW = nw; return true;
    }
    return false;
}

// Hopping disorder in the patch layout: the shapes with a table variant — 2 x 6 and 2 x 4 patches on one wavefront (L = 24, 20; 4 x 4 exists for the
// A/B), 2 x 2 patches on two to five (L = 22, 26, 34; disordered 28, 30 and 32); the others (L = 18, 36 and beyond) keep the generic kernels when the
// hopping is disordered.
inline bool patch_takes_disorder(int px, int py, int nw) {
    if (nw <= 1) return (px == 4 && py == 4) || (px == 2 && (py == 6 || py == 4));
    return px == 2 && py == 2 && nw >= 2 && nw <= 5;      // (12 entries per thread: 24 / 36 / 48 / 60 KB; four wavefronts: disordered 28 x 28, 30 x 30 and 32 x 32, elph_api.hip: detect_square; five: 34 x 34)
}

// The patch shape for an L x L lattice (0: none — the lattice keeps the generic kernels).
inline bool pick_patch(int L, int *PX, int *PY) {
    switch (L) {
        case 32: case 28: *PX = 4; *PY = 4; return true;
        case 24: case 18: *PX = 2; *PY = 6; return true;
        case 20: *PX = 2; *PY = 4; return true;
        case 30: *PX = 2; *PY = 10; return true;     // 15 x 3 lanes, 20 registers per vector (round 5)
        case 36: *PX = 4; *PY = 6; return true;      // 9 x 6 lanes, 24 registers per vector (round 6: the largest patch a wavefront carries — 6 vectors of the recursion = 288 registers, the overflow in AGPRs)
        default: return false;
    }
}

}   // namespace pgrid
