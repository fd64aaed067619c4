// cg_wg.hip — workgroup-resident conjugate gradient: the whole un-preconditioned solve of M^T M x = b
// (IterativeSolvers.jl:239-314) in ONE launch, Krylov vectors in registers, for lattices with a 4-colour lane program.
//
// Why: the two-kernel iteration (cg_fast_impl.inc) streams r, p, x, z through HBM/L2 every iteration — HBM-bound in a large
// batch (0.6-0.7 of the 8 TB/s peak), launch-bound for the reference's real call shape (1-2 right-hand sides: 8 us per
// iteration for 0.6 us of work).  Here a right-hand side belongs to a TEAM of G workgroups of W wavefronts; a wavefront owns
// T consecutive tau-slices for the whole solve and keeps p (own slices + one halo slice each side) and exp(-dtau V) in registers,
// x and r in its LDS (or registers, by shape).  Nothing of the Krylov vectors goes back to memory between iterations: an
// iteration costs two checkerboard-sweep passes (in registers — the DPP forms of cg_wg_dev.h — or in the wave's private LDS
// slabs) plus ONE MEETING of the team: every workgroup publishes FOUR sums — p.z, r.z, z.z and the r.r of the current residual —
// and the boundary slices of z; alpha = r.r / p.z, r'.r' = r.r - 2 alpha r.z + alpha^2 z.z gives beta and the stop test without a
// second reduction, and the neighbouring workgroup's boundary slice of r' is its r minus alpha times its z (see the loop).
// A meeting is hierarchical: wave partials -> LDS -> s_barrier (inside the workgroup), then — only if the team has more
// than one workgroup — one 64-byte record per workgroup through L2: eight 8-byte {tag = iteration, half of an f64} granules
// written by ONE sc1 store each and polled with sc1 loads until all tags carry the iteration number
// (cdna_hip_programming.md, Guideline 16, form R2: the data is the flag; no fence).  The boundary slices of z travel as granules
// of the same kind.  Every wave of a team reduces the same records in the same order, so alpha, beta and the stop decision are
// bit-identical across the team and from run to run.  (A sharded solve, SHARD, meets on two levels: the workgroups of a rank, then
// the ranks through their mailboxes.  -DELPH_SHARD_TWO_MEETINGS keeps round 2's two-meeting iteration for A/B.)
//
// Placement: team members are blocks with equal blockIdx % 8 (they share an XCD and its L2 under the observed round-robin
// placement — speed only, never correctness).  The grid may hold more teams than the chip can keep resident: blocks are
// dispatched in index order, a team's members have neighbouring indices, so at most the eight teams at the dispatch
// frontier wait (spinning) for members that start when another team has finished its solve.  Every spin is bounded by a
// wall-clock limit; a wave that gives up raises `abort`, all others leave within 64 polls, and the host falls back to the
// two-kernel iteration.

#include <algorithm>
#include <type_traits>

#include "cg_fast_common.h"

#include "cg_wg_dev.h"
#include "pgrid_dev.h"

namespace wg {

#ifdef ELPH_WG_ARRIVE
__device__ unsigned long long g_wg_arrive[4096 * 4];
// (tools/diag_wg_timeline.py) per workgroup: wall clock at the top of its first iteration, at the end of its first iteration, when it saw
// `done`, and after its last stores — what a launch of few iterations spends outside them
__device__ unsigned long long g_wg_timeline[4096 * 4];
#define TL(k) do { if (threadIdx.x == 0 && blockIdx.x < 4096) g_wg_timeline[blockIdx.x * 4 + (k)] = (unsigned long long)wall_clock64(); } while (0)
#else
#define TL(k) do { } while (0)
#endif

// SQ: the DPP form for the 16 x 16 square lattice in the reference's colouring (NPL = 4, no LDS slabs; Holstein: uniform hopping in
// two scalars, disordered hopping in per-site registers; SSH: a table set per time slice, SqSsh); otherwise the lane-program form
//     FORM 0: lane program, 1: the square-lattice DPP form (SQ), 2: the honeycomb DPP form (HC: 12 x 12 cells, six sites per lane of
//     which a quarter are mirror lanes, uniform hopping; cg_wg_dev.h), 4: the 8 x 8 DPP form, 5: the GRID form — any other even-L
//     square lattice up to 16 x 16 (cg_fast_common.h: 2 x 2 patches on a G x G grid of lanes, crossings by ds_bpermute), 6: the HGRID
//     form — any other honeycomb lattice up to 16 x 16 cells (1, 2 or 4 cells per lane on a grid of lanes; m.hc_L cells per side)
// SHARD: this launch is one rank's part of a solve over several GPUs (T = 1, lane-program form)
// X0Z: the initial guess is known to be zero (the library zeroed it for this solve): x0 is not read
// RANKS (with SHARD): SEVERAL RANKS OF ONE SHARDED SOLVE IN ONE LAUNCH (the slabs of a lattice beyond one wave's slice on ONE GPU,
// slabs.hip): rank q's G workgroups are the blocks q G .. q G + G - 1 and every rank's launch arguments — its slab's buffers, tables,
// control block, mailboxes — come from memory (R.ranks).  The ranks wait for each other, so they must all be resident: one grid
// guarantees what P streams do not (streams share the process's few hardware queues, and a queue runs its launches one after the other).
struct WgRankArgs { CgBufs B; ModelDev m; WgCtl R; ShardCtl Sh; };
template <int NPL, int T, bool SSH, bool UNI, int FORM, bool SHARD, bool X0Z = false, bool RANKS = false>
__global__ void __launch_bounds__(512) k_cg_wg(CgBufs B, ModelDev m, WgCtl R, ShardCtl Sh) {
    static_assert(!RANKS || SHARD, "several ranks per launch: sharded solves only");
    unsigned bid = 0;
    if constexpr (RANKS) {
        const WgRankArgs *__restrict__ A = static_cast<const WgRankArgs *>(R.ranks);
        const unsigned rk = blockIdx.x / (unsigned)R.G;
        bid = blockIdx.x - rk * (unsigned)R.G;
        B = A[rk].B; m = A[rk].m; Sh = A[rk].Sh; R = A[rk].R;
    }
    constexpr bool SQ = FORM == 1, HC = FORM == 2, S8 = FORM == 4, TG = FORM == 7, GR = FORM == 5 || TG, HG = FORM == 6, REGX = FORM != 0;      // REGX: the checkerboard exchanges registers, no LDS slabs
    // (FORM 7, TG: an even-L TRIANGULAR lattice up to 16 x 16 in the GRID layout — the same 2 x 2 patches, the two diagonal colours of
    //  pgrid::Tri<2, 2> added to the sweep, c^6 where the square lattice takes c^4)
    static_assert(!TG || !SHARD, "triangular grid form: no shards");
    static_assert(!HG || ((NPL == 2 || NPL == 4 || NPL == 8) && UNI && !SSH && T <= 2), "honeycomb grid form: 1, 2 or 4 cells per lane, uniform hopping, at most two slices per wave");
    static_assert(!GR || (NPL == 4 && UNI && !SSH && T <= 4), "grid form (even-L square lattices, 2 x 2 patches on a G x G lane grid): uniform hopping, at most four slices per wave");
    static_assert(!S8 || (NPL == 1 && UNI && !SSH && !SHARD), "8 x 8 DPP form: one site per lane, uniform hopping");
    static_assert(!SHARD || (T == 1 && (FORM == 0 || FORM == 5 || FORM == 6)), "sharded solves: one slice per wave; lane-program, GRID or HGRID form (the slab closed into a ring)");
    static_assert(!HC || (NPL == HC_NPL && UNI && !SSH && T <= 3), "honeycomb DPP form: six sites per lane, uniform hopping");
    extern __shared__ __attribute__((aligned(16))) double lds[];
    static_assert(!SQ || NPL == 4, "DPP form: the 16 x 16 square lattice, four sites per lane");
    static_assert(!(SQ && SSH) || (!UNI && T <= 2), "DPP form with bond phonons: a table set per slice, at most two slices per wave");
    static_assert(!SQ || UNI || T <= 2, "DPP form with per-site hopping: 32 registers of (cosh, sinh) leave room for two slices");
    constexpr int NE = MC * ((NPL + 1) / 2);
    constexpr int HS = NPL * WAVE, SL = slab_len<NPL>();
    // slices in LDS: LSL lanes per register, HSL doubles per slice.  Honeycomb form: the 48 REAL lanes only — a mirror lane READS the
    // slot of the lane it mirrors (lr) and writes nothing (lwok): its r, exp(-dtau V) and halo values are the real lane's by
    // construction, and a quarter of the LDS is not spent on copies (what lets 3 slices per wave fit)
    constexpr int LSL = HC ? 48 : WAVE, HSL = NPL * LSL;
    constexpr int NSLAB = REGX ? 0 : T + 1;            // LDS slabs per wave (lane-program form)
    constexpr int NT = SSH ? T + 1 : 1;                // hopping-table sets (SSH: one per slice t0 .. t0+T)
    constexpr int NEJ = SSH ? 1 : T + 1;               // exp(-dtau V) slices (SSH: exp(dtau mu), per site only)
    // where the loop-invariant and the update-only vectors live.  4 slices per wave (DPP form, throughput shape): exp(-dtau V) moves
    // to LDS (read twice per iteration; 40 registers) and x stays in registers instead (LDS is full); otherwise E in registers
    // (0.35 us per iteration faster at 2 slices per wave), x and r in LDS
    // bond phonons, DPP form, 2 slices per wave: the table set of slice t0 in LDS (24 doubles per lane), x in registers
    constexpr bool S_LDS = SQ && SSH && T == 2;
    // honeycomb DPP form, 2 slices per wave: exp(-dtau V) in LDS too (six sites per lane: 36 registers)
    constexpr bool E_LDS = ((SQ || GR) && T >= 4) || (HC && T >= 2), X_REG = ((SQ || GR) && T >= 4) || S_LDS || (HC && T >= 3);
    // (honeycomb: neighbouring waves SHARE the slice of exp(-dtau V) between them — [W T + 1] slices per workgroup instead of W (T + 1);
    //  both write the same values to it)
    constexpr bool E_SHARED = HC;
    // honeycomb, 3 slices per wave: x lives in MEMORY (it is touched once per iteration, x += alpha p): its slices are loaded behind the
    // sums — the meeting that follows hides the round trip — and stored back by the update; between two updates no register holds it
    constexpr bool X_GLB = HC && T >= 3;
    // honeycomb, 3 slices per wave: the two HALO slices of p wait in LDS between the p-update that makes them and the mat-vec that
    // consumes them (24 registers that would otherwise sit through both sweeps)
    constexpr bool PH_LDS = HC && T >= 3;
    constexpr int NSREG = (SQ && SSH) ? (S_LDS ? T : T + 1) : 1;
    // ONE: the single-meeting iteration (see the loop) — for a shard too: its meeting is two-level (workgroups of the rank, then ranks)
    // and carries the ghost rows of z
#ifdef ELPH_SHARD_TWO_MEETINGS
    constexpr bool ONE = !SHARD;                       // (A/B build: a shard keeps round 2's two-meeting iteration)
#else
    constexpr bool ONE = true;
#endif
    const int W = R.W, G = R.G;
    // (a shard: ONE right-hand side, its G workgroups are the whole grid, spread over the XCDs — several ranks on one GPU, the test
    //  box, would otherwise pile their teams onto the XCD where every dispatch starts: 2 x 20 workgroups do not fit its 32 CUs)
    const int xcd = SHARD ? 0 : (blockIdx.x & 7), idx = SHARD ? (RANKS ? bid : blockIdx.x) : (blockIdx.x >> 3);
    const int tq = idx / G, g = idx - tq * G;
    // PERSISTENT TEAMS (-DELPH_WG_PERSISTENT; not the default): the grid holds at most as many teams as the chip keeps resident
    // (R.teams_per_xcd per XCD, one workgroup per CU) and a team takes the right-hand sides tq, tq + teams, ... of its XCD's residue class
    // one after the other, so that no workgroup ever waits for one that has not been dispatched yet.  The default keeps ONE solve per
    // workgroup and an oversubscribed grid (teams at the dispatch frontier wait for members that start when another team ends —
    // blocks are dispatched in index order; the wall-clock bound, the fallback and the cool-down of elph_wg_cg are the guard):
    // measured, the loop around the solve costs 4 % (30.7 against 29.5 us per iteration of 288 right-hand sides, 5.51 against 5.22 at
    // 48 — what stays live across the loop spills 126 scalar registers), profiles/r03/wg_persistent_teams.log.
#ifdef ELPH_WG_PERSISTENT
    for (int tqi = tq;; tqi += R.teams_per_xcd) {
    // (the thread number is laundered per right-hand side: everything derived from it — sites, LDS offsets, DPP partners, addresses —
    //  is made afresh for each solve; hoisted out of this loop it would sit in registers for all of them)
    int tid = (int)threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & (WAVE - 1);
#else
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & (WAVE - 1);
    const int tqi = tq;
    {
#endif
    const int rhs = tqi * 8 + xcd;
#ifdef ELPH_WG_ARRIVE
    // diagnostic build (tools/diag_wg_arrive.py): where and when every workgroup of the grid started — XCC_ID and the hardware id of its
    // CU, the wall clock — to see which members a team that timed out was waiting for
#ifndef ELPH_WG_ARRIVE_NOSTART
    if (threadIdx.x == 0 && blockIdx.x < 4096) {
        g_wg_arrive[blockIdx.x * 4 + 0] = 1ull + (unsigned long long)__builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20);      // XCC_ID[3:0]
        g_wg_arrive[blockIdx.x * 4 + 1] = (unsigned long long)__builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4);            // HW_ID
        g_wg_arrive[blockIdx.x * 4 + 2] = (unsigned long long)wall_clock64();
    }
#endif
#endif
    if (rhs >= B.nrhs) return;
    const int N = m.N, L = m.L;
    const int t0 = (g * W + wv) * T;
    const int lr = HC ? 12 * (hc_src_lane(lane) >> 4) + (hc_src_lane(lane) & 15) - 2 : lane;      // LDS slot this lane reads (and, if lwok, writes)
    const int GGX = GR ? m.grid_GX : 0, GGY = GR ? m.grid_GY : 0;      // grid form: GX x GY lanes hold the lattice, the rest idle
    const int HLX = HG ? m.hc_LX : 0, HLY = HG ? m.hc_LY : 0;         // honeycomb grid form: (LX / PX) x (LY / PY) lanes
    const bool lwok = HC ? hc_real(lane) : (GR ? lane < GGX * GGY : (HG ? lane < (HLX / HgDim<NPL>::PX) * (HLY / HgDim<NPL>::PY) : true));
    const size_t ndim = (size_t)N * L;
    double *slab = lds + (size_t)wv * NSLAB * SL;
    double *rall = lds + (size_t)W * NSLAB * SL;       // [W][T][HS]: r of every wave's slices — neighbours read their halo slices here
    double *rl = rall + (size_t)wv * T * HSL;
    double *xl = rall + (size_t)W * T * HSL + (size_t)wv * T * HSL;    // [W][T][HSL]: this wave's slices of x (unless X_REG)
    double *eall = rall + (size_t)(X_REG ? 1 : 2) * W * T * HSL;       // [W][T+1][HSL]: exp(-dtau V) of slices t0 .. t0+T (E_LDS)
    double *el = eall + (size_t)wv * (E_SHARED ? T : T + 1) * HSL;
    double *partA = eall + (E_LDS ? (E_SHARED ? ((size_t)W * T + 1) * HSL : (size_t)W * (T + 1) * HSL) : (S_LDS ? (size_t)W * SQ_TABS * WAVE : 0)), *partB = partA + 8, *bc = partA + 16;   // bc: p.z total, r.r total, 0.0 = a poller gave up
    // single-meeting form: part[4][8] wave partials of p.z, r.z, z.z, r.r | tot[8]: the four totals, [4] the direct r.r of the fallback
    // meeting, [5] 0.0 = a poller gave up | partF[8] | rhalo[2][HS]: the boundary slices of r of the two neighbouring workgroups (kept in
    // LDS rather than in registers: the 4-slice shape has none to spare)
    double *part = partA, *tot = partA + 32, *partF = partA + 40, *rhalo = partA + 48, *zhalo = rhalo + 2 * HSL;
    double *phl = zhalo + 2 * HSL + (size_t)wv * 2 * HSL;             // [W][2][HSL]: the halo slices of p (PH_LDS)
#define PHALO(k, q) phl[((k) ? 1 : 0) * HSL + lr + (q) * LSL]   // zhalo[2][HS]: their boundary slices of z
    auto wrap = [L](int t) { return (t < 0) ? t + L : ((t >= L) ? t - L : t); };
    auto sgn = [](int t) { return (t == 0) ? -1.0 : 1.0; };
    const CgParams P = B.params;
    CgState *st2 = B.state + 2 * rhs;
    const CgState S = ld_state(st2);
#ifdef ELPH_WG_PERSISTENT
    if (!SHARD && (S.done || S.seq != 0)) continue;    // fresh solves only (the host guarantees it); a shard seeds its state below
#else
    if (!SHARD && (S.done || S.seq != 0)) return;      // fresh solves only (the host guarantees it); a shard seeds its state below
#endif

    // site of register q of this lane: lane + 64 q (layout S order), or the column segments of the DPP form
    int sc[NPL], ss[NPL];
    bool live[NPL], own[NPL];                          // own: the site enters the inner products (a shard counts its own rows only)
#pragma unroll
    for (int q = 0; q < NPL; ++q) {
        const int s = SQ ? sq_patch_site(lane, q) : (HC ? hc_site(lane, q) : (S8 ? s8_site(lane) : (GR ? grid_site(lane, q, GGX, GGY) : (HG ? hgrid_site<NPL>(lane, q, HLX, HLY) : lane + q * WAVE))));
        live[q] = (GR || HG) ? lwok : (REGX || s < N);            // (DPP forms: every register of every lane holds a site — no selects in the sums)
        own[q] = SHARD ? (live[q] && s >= Sh.own_lo && s < Sh.own_hi) : (HC ? hc_real(lane) : live[q]);     // (honeycomb: mirror lanes carry copies)
        sc[q] = live[q] ? s : N - 1;
        ss[q] = live[q] ? s : N;                          // (a shard asks where a site sits among own and ghost rows: an idle register sits nowhere)
    }
    double *xg = B.x + (size_t)rhs * ndim, *rg = B.r + (size_t)rhs * ndim;
    const double *Ech = m.E + (size_t)(rhs % m.nchains) * m.E_chain_stride;

    // x and r live in this wave's LDS (lane-linear, conflict-free): both are only touched by the two vector updates, never by the
    // mat-vec — no register across the mat-vec, no vector-memory traffic inside the loop that a meeting's poll would wait behind,
    // and the neighbouring waves read their halo slices of r straight from here
    double zw[T + 1][NPL], p[T + 2][NPL], E[E_LDS ? 1 : NEJ][NPL], xr[X_REG ? T : 1][NPL];
    // (x0 = 0 known to the library: a compile-time variant, X0Z, does not read it — 12 us less per launch of 288 right-hand sides.  As a
    //  RUN-TIME choice, and likewise reading the own slices of r0 once for r and p0 = r0, it changes the register assignment of the
    //  LOOP — 256 registers, none to spare at 4 slices per wave — and costs it 8-20 %: 5.6-6.4 against 5.2 us per iteration at 48
    //  right-hand sides (profiles/r03/wg_load_variants.log).)
#pragma unroll
    for (int j = 0; j < T; ++j)
#pragma unroll
        for (int q = 0; q < NPL; ++q) {
            if (lwok) rl[j * HSL + lr + q * LSL] = rg[(size_t)(t0 + j) * N + sc[q]];
            if (X_GLB) {}      // (x0 is where it is: the caller's initial guess in memory)
            else if (X_REG) xr[X_REG ? j : 0][q] = X0Z ? 0.0 : xg[(size_t)(t0 + j) * N + sc[q]];
            else if (lwok) xl[j * HSL + lr + q * LSL] = X0Z ? 0.0 : xg[(size_t)(t0 + j) * N + sc[q]];
        }
#pragma unroll
    for (int j = 0; j < T + 2; ++j)
#pragma unroll
        for (int q = 0; q < NPL; ++q) {
            p[j][q] = rg[(size_t)wrap(t0 + j - 1) * N + sc[q]];      // p0 = r0 (:272): one vector less to read
            if (PH_LDS && (j == 0 || j == T + 1) && lwok) PHALO(j, q) = p[j][q];
        }
#pragma unroll
    for (int j = 0; j < NEJ; ++j)
#pragma unroll
        for (int q = 0; q < NPL; ++q) {
            const double ev = Ech[(size_t)wrap(t0 + j) * m.E_tau_stride + sc[q]];
            if (E_LDS) { if (lwok) el[j * HSL + lr + q * LSL] = ev; }
            else E[E_LDS ? 0 : j][q] = ev;
        }
    unsigned ij[NE];
    Tab<NE, UNI> tab[NT];
    SqCtx<UNI> X;
    SqSsh<NSREG, S_LDS> XS;
    HcCtx XH;
    S8Ctx X8;
    GridCtx XG;
    HgCtx XHG;
    pgrid::Ctx XT;
    if constexpr (GR) {
        XG = grid_ctx(lane, GGX, GGY, m.c_uni, m.s_uni);
        if constexpr (TG) { XT = pgrid::Tri<2, 2>::make_ctx(lane, 2 * GGX, m.c_uni, m.s_uni); XG.k4 = XT.ks; }      // (k4 carries c^6 here)
    } else if constexpr (HG) {
        XHG = hgrid_ctx<NPL>(lane, HLX, HLY, m.c_uni, m.s_uni);
    } else if constexpr (S8) {
        X8.th = m.s_uni / m.c_uni; X8.k4 = (m.c_uni * m.c_uni) * (m.c_uni * m.c_uni);
        X8.yx = sq_patch_ycross(lane); X8.xodd = (lane >> 1) & 1;
    } else if constexpr (HC) {
        XH.th = m.s_uni / m.c_uni; XH.k3 = m.c_uni * m.c_uni * m.c_uni;
        XH.up = (lane + 16) & (WAVE - 1); XH.dn = (lane + 48) & (WAVE - 1);
    } else if constexpr (SQ && SSH) {
        // table set j = the hopping of slice t0 + j, gathered from the per-(tau, bond) tables through the site -> bond map of each colour
        ssh_chain_select(m, rhs);
        XS.yx = sq_patch_ycross(lane);
        XS.l0 = eall + (size_t)wv * SQ_TABS * WAVE + lane;
#pragma unroll
        for (int j = 0; j <= T; ++j) {
            const double *cj = m.c + (size_t)wrap(t0 + j) * m.cs_tau_stride, *sj = m.s + (size_t)wrap(t0 + j) * m.cs_tau_stride;
            SqTabS tb;
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                const int b0 = m.sq_bond[0 * N + sc[2 * pr]], b2 = m.sq_bond[2 * N + sc[pr]];
                tb.ci[0][pr] = cj[b0]; tb.si[0][pr] = sj[b0];
                tb.ci[1][pr] = cj[b2]; tb.si[1][pr] = sj[b2];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int b1 = m.sq_bond[1 * N + sc[k]], b3 = m.sq_bond[3 * N + sc[k]];
                tb.cx[0][k] = cj[b1]; tb.sx[0][k] = sj[b1];
                tb.cx[1][k] = cj[b3]; tb.sx[1][k] = sj[b3];
            }
            if (S_LDS && j == 0) {
                double *d = eall + (size_t)wv * SQ_TABS * WAVE + lane;
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    d[(0 + pr) * WAVE] = tb.ci[0][pr]; d[(2 + pr) * WAVE] = tb.si[0][pr];
                    d[(4 + pr) * WAVE] = tb.ci[1][pr]; d[(6 + pr) * WAVE] = tb.si[1][pr];
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    d[(8 + k) * WAVE] = tb.cx[0][k]; d[(12 + k) * WAVE] = tb.sx[0][k];
                    d[(16 + k) * WAVE] = tb.cx[1][k]; d[(20 + k) * WAVE] = tb.sx[1][k];
                }
            } else {
                XS.t[S_LDS ? (j > 0 ? j - 1 : 0) : (j < NSREG ? j : 0)] = tb;
            }
        }
    } else if constexpr (SQ) {
        X.yx = sq_patch_ycross(lane);
        if constexpr (UNI) {
            X.c[0][0] = m.c_uni; X.s[0][0] = m.s_uni / m.c_uni; X.k4 = (m.c_uni * m.c_uni) * (m.c_uni * m.c_uni);
        } else {
            X.k4 = 1.0;
#pragma unroll
            for (int col = 0; col < 4; ++col)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int bd = m.sq_bond[col * N + sc[k]];
                    X.c[UNI ? 0 : col][UNI ? 0 : k] = m.c[bd];
                    X.s[UNI ? 0 : col][UNI ? 0 : k] = m.s[bd];
                }
        }
    } else {
#pragma unroll
        for (int e = 0; e < NE; ++e) ij[e] = m.lp_ij[e * WAVE + lane];
        if (SSH) {
            ssh_chain_select(m, rhs);
#pragma unroll
            for (int j = 0; j < NT; ++j)
                load_tab<NE, UNI>(tab[j], m.lp_c + (size_t)wrap(t0 + j) * m.lp_tau_stride, m.lp_s + (size_t)wrap(t0 + j) * m.lp_tau_stride, lane, m);
        } else {
            load_tab<NE, UNI>(tab[0], m.lp_c, m.lp_s, lane, m);
        }
    }
#define EXPV(j, q) (E_LDS ? el[(j) * HSL + lr + (q) * LSL] : E[(SSH || E_LDS) ? 0 : (j)][q])

    // records and boundary granules exist TWICE, by the parity of the iteration: a workgroup that has met for iteration k goes on and
    // publishes its record of k + 1 one mat-vec later — into the OTHER set, so that a member that is late reading the records of k
    // (its polling wave held up: two workgroups sharing a CU, a time-sliced GPU) still finds them; nobody can reach k + 2 before that
    // member has published k + 1, i.e. finished reading k.  (With ONE set such a member waited for tags that had already moved on —
    // the time-out of the experimental 4-wave shape, tools/diag_wg_arrive.py.)
    constexpr size_t SLOTS_RHS = ONE ? SLOTS_PER_RHS : 2 * 64;
    u64 *const slots0 = R.slots + (size_t)rhs * 2 * SLOTS_RHS;
    u64 *const bnd0 = R.bnd + (size_t)rhs * 2 * G * 2 * HS * 2;     // [parity][G][first | last slice][HS][2 granules]
    const int gm = (g == 0) ? G - 1 : g - 1, gp = (g == G - 1) ? 0 : g + 1;
    double rho = S.rho, kmin = S.kmin, eps = S.eps;
    double eps0 = S.eps0, normb = S.normb;

    // (ordered before their first readers by the barriers that follow.  bc — the seed meeting of a shard, the two-meeting form — shares
    //  its three words with part[16..18], the z.z partials of waves 0..2: in the single-meeting form of an un-sharded solve it must NOT
    //  be touched here — nothing orders this store of wave 0 before the first iteration's part[] stores of a wave that finished its
    //  set-up sooner, and a late 1.0 in place of wave 2's z.z partial costs that solve its first beta: one iteration more, about one
    //  first solve in three of a fresh handle with the slow set-up of the bond-phonon DPP form)
    if (threadIdx.x == 0) { tot[5] = 1.0; if (SHARD || !ONE) bc[2] = 1.0; }
    // single-meeting form: the boundary waves of a workgroup keep the neighbouring workgroup's boundary slice of r (p0 = r0: it sits in
    // the halo of p)
    if constexpr (ONE) {
        if (G > 1) {
            if (wv == 0) {
#pragma unroll
                for (int q = 0; q < NPL; ++q) if (lwok) rhalo[lr + q * LSL] = p[0][q];
            }
            if (wv == W - 1) {
#pragma unroll
                for (int q = 0; q < NPL; ++q) if (lwok) rhalo[HSL + lr + q * LSL] = p[T + 1][q];
            }
        }
    }
    // ghost sites of this lane: where their values arrive in the own mailbox (nullptr: not a ghost site), slice t0
    const u64 *gaddr[NPL];
    if constexpr (SHARD) {
#pragma unroll
        for (int q = 0; q < NPL; ++q) {
            const int s = ss[q];
            gaddr[q] = (s < Sh.own_lo) ? sh_ghost(Sh.mail[Sh.rank], 0, L, Sh.cap_ghost, t0, s)
                     : (s >= Sh.own_hi && s < N) ? sh_ghost(Sh.mail[Sh.rank], 1, L, Sh.cap_ghost, t0, s - Sh.own_hi) : nullptr;
        }
        // x0 = 0, r0 = p0 = b (the host put b into r and p): |b|^2 over the own sites of all ranks seeds the state
        // (IterativeSolvers.jl:259-274 with x = 0: eps0 = 1, rho0 = |b|^2)
        double a0 = 0.0;
#pragma unroll
        for (int q = 0; q < NPL; ++q) if (own[q]) a0 += p[1][q] * p[1][q];
        a0 = wave_sum_dpp(a0);
        if (lane == 0) partA[wv] = a0;
        wg_barrier();
        if (wv == 0) {
            const double mine = wg_sum(partA, W, lane);
            sh_publish<RANKS>(Sh, 0, g, G, mine, 1u, lane);
            const u64 *none[NPL];
#pragma unroll
            for (int q = 0; q < NPL; ++q) none[q] = nullptr;
            double tot = 0.0, dummy[NPL];
            const bool ok = sh_poll<NPL, RANKS>(Sh, true, 0, G, none, 1u, lane, R, tot, dummy);
            if (lane == 0) { bc[0] = tot; if (!ok) bc[2] = 0.0; }
        }
        wg_barrier();
        if (bc[2] == 0.0) return;
        const double bb = bc[0];
        normb = sqrt(bb); eps0 = 1.0; eps = 1.0; rho = bb; kmin = 0.0;
        wg_barrier();                                   // bc[0] is rewritten by the first meeting of the loop
    }
    // screens of the stop test (see there)
    const double rr_far = (P.tol * normb) * (P.tol * normb) * 1.000001, y_num = 4.0 * (eps0 * normb) * (eps0 * normb);
    const double it_kappa = 0.17 * sqrt(P.kmax);
    STAMP_DECL;
    for (long long seq = 0;; ++seq) {
        const unsigned epoch = R.epoch0 + (unsigned)seq + (SHARD ? 2u : 1u);
        const unsigned par = epoch & 1u;
        u64 *const slotsA = slots0 + (size_t)par * SLOTS_RHS, *const slotsB = slotsA + (ONE ? SLOTS_A : 64);
        u64 *const bnd = bnd0 + (size_t)par * G * 2 * HS * 2;
        const size_t ghp = SHARD ? (size_t)par * 2 * L * Sh.cap_ghost * 2 : 0;      // (a shard: the ghost rows of z in the mailboxes, by parity too)
        STAMP(9);
        if (seq == 0) TL(0);
        if (seq == 1) TL(1);
        // ---- z = M^T M p on the own slices:  w(t) = p(t) - sg(t) CB_t [E(t) p(t-1)]  for t = t0 .. t0+T  (T+1 forward sweeps at once),
        //      z(t) = w(t) - sg(t+1) E(t+1) CB_{t+1}^T w(t+1)  for t = t0 .. t0+T-1  (T reverse sweeps at once)
        // (w and z share registers: z(t0+j) overwrites w(t0+j) once the reverse sweep of w(t0+j+1) has been taken)
        double (&w)[T + 1][NPL] = zw;
        if constexpr (S8) {
#pragma unroll
            for (int k = 0; k <= T; ++k) w[k][0] = EXPV(k, 0) * p[k][0];
            s8_sweepN<T + 1, false>(w, X8);
#pragma unroll
            for (int k = 0; k <= T; ++k) w[k][0] = p[k + 1][0] - sgn(wrap(t0 + k)) * X8.k4 * w[k][0];
            double gq[T][1];
#pragma unroll
            for (int i = 0; i < T; ++i) gq[i][0] = w[i + 1][0];
            s8_sweepN<T, true>(gq, X8);
#pragma unroll
            for (int i = 0; i < T; ++i) w[i][0] = w[i][0] - sgn(wrap(t0 + i + 1)) * X8.k4 * (EXPV(i + 1, 0) * gq[i][0]);      // z(t0+i)
        } else if constexpr (HC) {
#pragma unroll
            for (int k = 0; k <= T; ++k)
#pragma unroll
                for (int q = 0; q < NPL; ++q) w[k][q] = EXPV(k, q) * ((PH_LDS && k == 0) ? PHALO(0, q) : p[k][q]);
            hc_sweepN<T + 1, false>(w, XH);
#pragma unroll
            for (int k = 0; k <= T; ++k) {
                const double sg = sgn(wrap(t0 + k)) * XH.k3;
#pragma unroll
                for (int q = 0; q < NPL; ++q) w[k][q] = ((PH_LDS && k == T) ? PHALO(1, q) : p[k + 1][q]) - sg * w[k][q];
            }
            // reverse sweeps of the slabs J0 .. J0 + NB - 1 at once (both compile-time: a run-time index into w would put it on the stack)
            auto rev = [&](auto nb, auto j0c) __attribute__((always_inline)) {
                constexpr int NB = decltype(nb)::value, j0 = decltype(j0c)::value;
                double gq[NB][NPL];
#pragma unroll
                for (int i = 0; i < NB; ++i)
#pragma unroll
                    for (int q = 0; q < NPL; ++q) gq[i][q] = w[j0 + i + 1][q];
                hc_sweepN<NB, true>(gq, XH);
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    const double sg = sgn(wrap(t0 + j0 + i + 1)) * XH.k3;
#pragma unroll
                    for (int q = 0; q < NPL; ++q) w[j0 + i][q] = w[j0 + i][q] - sg * (EXPV(j0 + i + 1, q) * gq[i][q]);     // z(t0+j0+i)
                }
            };
            using std::integral_constant;
            if constexpr (T == 3) { rev(integral_constant<int, 2>(), integral_constant<int, 0>()); rev(integral_constant<int, 1>(), integral_constant<int, 2>()); }
            else rev(integral_constant<int, T>(), integral_constant<int, 0>());
        } else if constexpr (HG) {
#pragma unroll
            for (int k = 0; k <= T; ++k)
#pragma unroll
                for (int q = 0; q < NPL; ++q) w[k][q] = EXPV(k, q) * p[k][q];
            hgrid_sweepN<NPL, T + 1, false>(w, XHG);
#pragma unroll
            for (int k = 0; k <= T; ++k) {
                const double sg = sgn(wrap(t0 + k)) * XHG.k3;
#pragma unroll
                for (int q = 0; q < NPL; ++q) w[k][q] = p[k + 1][q] - sg * w[k][q];
            }
            double gq[T][NPL];
#pragma unroll
            for (int i = 0; i < T; ++i)
#pragma unroll
                for (int q = 0; q < NPL; ++q) gq[i][q] = w[i + 1][q];
            hgrid_sweepN<NPL, T, true>(gq, XHG);
#pragma unroll
            for (int i = 0; i < T; ++i) {
                const double sg = sgn(wrap(t0 + i + 1)) * XHG.k3;
#pragma unroll
                for (int q = 0; q < NPL; ++q) w[i][q] = w[i][q] - sg * (EXPV(i + 1, q) * gq[i][q]);      // z(t0+i)
            }
        } else if constexpr (GR) {
#pragma unroll
            for (int k = 0; k <= T; ++k)
#pragma unroll
                for (int q = 0; q < 4; ++q) w[k][q] = EXPV(k, q) * p[k][q];
            if constexpr (TG) {
#pragma unroll
                for (int k = 0; k <= T; ++k) pgrid::Tri<2, 2>::apply<false>(w[k], XT);
            } else grid_sweepN<T + 1, false, (T >= 4) ? 1 : 2>(w, XG);
#pragma unroll
            for (int k = 0; k <= T; ++k) {
                const double sg = sgn(wrap(t0 + k)) * XG.k4;
#pragma unroll
                for (int q = 0; q < 4; ++q) w[k][q] = p[k + 1][q] - sg * w[k][q];
            }
            constexpr int RBG = (T >= 4) ? 2 : T;            // reverse sweeps per batch (4 slices per wave: two batches, as in the DPP form)
#pragma unroll
            for (int j0 = 0; j0 < T; j0 += RBG) {
                double gq[RBG][4];
#pragma unroll
                for (int i = 0; i < RBG; ++i)
#pragma unroll
                    for (int q = 0; q < 4; ++q) gq[i][q] = w[j0 + i + 1][q];
                if constexpr (TG) {
#pragma unroll
                    for (int i = 0; i < RBG; ++i) pgrid::Tri<2, 2>::apply<true>(gq[i], XT);
                } else grid_sweepN<RBG, true, (T >= 4) ? 1 : 2>(gq, XG);
#pragma unroll
                for (int i = 0; i < RBG; ++i) {
                    const double sg = sgn(wrap(t0 + j0 + i + 1)) * XG.k4;
#pragma unroll
                    for (int q = 0; q < 4; ++q) w[j0 + i][q] = w[j0 + i][q] - sg * (EXPV(j0 + i + 1, q) * gq[i][q]);      // z(t0+j0+i)
                }
            }
        } else if constexpr (SQ) {
#pragma unroll
            for (int k = 0; k <= T; ++k)
#pragma unroll
                for (int q = 0; q < 4; ++q) w[k][q] = EXPV(k, q) * p[k][q];
            if constexpr (SSH) sq_sweepS<T + 1, false, 0>(w, XS);
            else sq_sweepN<T + 1, false, UNI>(w, X);
#pragma unroll
            for (int k = 0; k <= T; ++k) {
                const double sg = UNI ? sgn(wrap(t0 + k)) * X.k4 : sgn(wrap(t0 + k));
#pragma unroll
                for (int q = 0; q < 4; ++q) w[k][q] = p[k + 1][q] - sg * w[k][q];
            }
            constexpr int RB = (T >= 4) ? 2 : T;             // reverse sweeps per batch (4 slices per wave: two batches, 16 registers less)
#pragma unroll
            for (int j0 = 0; j0 < T; j0 += RB) {
                double gq[RB][4];
#pragma unroll
                for (int i = 0; i < RB; ++i)
#pragma unroll
                    for (int q = 0; q < 4; ++q) gq[i][q] = w[j0 + i + 1][q];
                if constexpr (SSH) sq_sweepS<RB, true, 1>(gq, XS);          // (RB = T: one batch, the sets of t0+1 ..)
                else sq_sweepN<RB, true, UNI>(gq, X);
#pragma unroll
                for (int i = 0; i < RB; ++i) {
                    const double sg = UNI ? sgn(wrap(t0 + j0 + i + 1)) * X.k4 : sgn(wrap(t0 + j0 + i + 1));
#pragma unroll
                    for (int q = 0; q < 4; ++q) w[j0 + i][q] = w[j0 + i][q] - sg * (EXPV(j0 + i + 1, q) * gq[i][q]);    // z(t0+j0+i)
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k <= T; ++k)
#pragma unroll
                for (int q = 0; q < NPL; ++q) slab[k * SL + lane + q * WAVE] = EXPV(k, q) * p[k][q];
            WAVE_LDS_ORDER();
            sweepN<NPL, T + 1, false, UNI, NT, SSH ? 1 : 0, 0>(slab, ij, tab, m.ncol);
#pragma unroll
            for (int k = 0; k <= T; ++k) {
                const double sg = sgn(wrap(t0 + k));
#pragma unroll
                for (int q = 0; q < NPL; ++q) w[k][q] = p[k + 1][q] - sg * slab[k * SL + lane + q * WAVE];
            }
            WAVE_LDS_ORDER();
#pragma unroll
            for (int j = 0; j < T; ++j)
#pragma unroll
                for (int q = 0; q < NPL; ++q) slab[j * SL + lane + q * WAVE] = w[j + 1][q];
            WAVE_LDS_ORDER();
            sweepN<NPL, T, true, UNI, NT, SSH ? 1 : 0, SSH ? 1 : 0>(slab, ij, tab, m.ncol);
#pragma unroll
            for (int j = 0; j < T; ++j) {
                const double sg = sgn(wrap(t0 + j + 1));
#pragma unroll
                for (int q = 0; q < NPL; ++q) w[j][q] = w[j][q] - sg * EXPV(j + 1, q) * slab[j * SL + lane + q * WAVE];       // z(t0+j)
            }
            WAVE_LDS_ORDER();
        }
        double (&z)[T + 1][NPL] = zw;                         // rows 0 .. T-1
        // (honeycomb, 3 slices per wave: the addresses of the boundary granules are made afresh every iteration from a laundered lane
        //  number — kept across the loop they are 2 registers each, 24 of them, in a kernel that has none to spare)
        int lane_b = lane;
        if constexpr (HC && T >= 3) asm volatile("" : "+v"(lane_b));
        double rr, hx[NPL];                                   // rr: r.r of the NEW residual; hx: the halo slice that comes from another workgroup (waves 0 and W-1)
        if constexpr (ONE) {
        // ================= single-meeting iteration ==========================================================================
        // The two meetings of the textbook iteration (p.z -> alpha; then r.r of the new residual and its boundary slices -> beta, halo
        // of p) fold into ONE: every workgroup publishes FOUR sums — p.z, r.z, z.z and the r.r of the current residual — and the
        // boundary slices of z.  With them every wave has  alpha = r.r / p.z  and, by the algebraic identity of r' = r - alpha z,
        //     r'.r' = r.r - 2 alpha r.z + alpha^2 z.z
        // so beta and the stop test need no second reduction, and the neighbour's boundary slice of r' is its r (kept here) minus
        // alpha times its z (just received).  The direct r.r enters every iteration afresh (summed from the actual vector when it is
        // made, published one meeting later): the identity is applied for ONE step only, its rounding error (a few ulp of r.r) never
        // accumulates.  When the step shrinks the residual so much that the identity cancels (r'.r' < r.r / 1000: never in the
        // hundreds of iterations of these matrices, but possible on a nearly diagonal one) the direct sum is taken in a second,
        // single-sum meeting — the same branch in every wave of the team, since all hold the same bits.
        double s_pz = 0.0, s_rz = 0.0, s_zz = 0.0, s_rr = 0.0;
#pragma unroll
        for (int j = 0; j < T; ++j)
#pragma unroll
            for (int q = 0; q < NPL; ++q)
                if (own[q]) {
                    const double rv = rl[j * HSL + lr + q * LSL];
                    s_pz += p[j + 1][q] * z[j][q];
                    s_rz += rv * z[j][q];
                    s_zz += z[j][q] * z[j][q];
                    s_rr += rv * rv;
                }
        {
            const double k4 = wave_sum4(s_pz, s_rz, s_zz, s_rr, lane);
            if (lane < 4) part[lane * 8 + wv] = k4;
        }
        if constexpr (X_GLB) {
#pragma unroll
            for (int j = 0; j < T; ++j)
#pragma unroll
                for (int q = 0; q < NPL; ++q) xr[X_REG ? j : 0][q] = xg[(size_t)(t0 + j) * N + sc[q]];
        }
        if constexpr (SHARD) {
            // the rows of z my rank neighbours hold as ghosts: straight into their mailboxes (device-initiated stores over xGMI),
            // self-tagged — with alpha from the meeting the neighbour makes its ghost rows of the new residual itself
            const int prev = (Sh.rank + Sh.P - 1) % Sh.P, next = (Sh.rank + 1) % Sh.P;
#pragma unroll
            for (int q = 0; q < NPL; ++q) {
                const int s = ss[q];
                const u64 bits = (u64)__double_as_longlong(z[0][q]), tag = (u64)epoch << 32;
                if (s >= Sh.own_lo && s < Sh.own_lo + Sh.n_to_prev) {              // bottom rows -> previous rank's ghosts above its own rows
                    u64 *d = sh_ghost(Sh.mail[prev], 1, L, Sh.cap_ghost, t0, s - Sh.own_lo) + ghp;
                    st_mail<RANKS>(d, tag | (bits & 0xFFFFFFFFull)); st_mail<RANKS>(d + 1, tag | (bits >> 32));
                }
                if (s >= Sh.own_hi - Sh.n_to_next && s < Sh.own_hi) {              // top rows -> next rank's ghosts below its own rows
                    u64 *d = sh_ghost(Sh.mail[next], 0, L, Sh.cap_ghost, t0, s - (Sh.own_hi - Sh.n_to_next)) + ghp;
                    st_mail<RANKS>(d, tag | (bits & 0xFFFFFFFFull)); st_mail<RANKS>(d + 1, tag | (bits >> 32));
                }
            }
        }
        if (G > 1) {                                          // boundary slices of z for the neighbouring workgroups (self-tagged granules)
            if (wv == 0) {
#pragma unroll
                for (int q = 0; q < NPL; ++q) {
                    st_f64_gran(bnd + (((size_t)g * 2 + 0) * HS + lane_b + q * WAVE) * 2, z[0][q], epoch);
                    if constexpr (HC && T >= 3) __builtin_amdgcn_sched_barrier(0);      // (one value's granules and address at a time)
                }
            }
            if (wv == W - 1) {
#pragma unroll
                for (int q = 0; q < NPL; ++q) {
                    st_f64_gran(bnd + (((size_t)g * 2 + 1) * HS + lane_b + q * WAVE) * 2, z[T - 1][q], epoch);
                    if constexpr (HC && T >= 3) __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        STAMP(0);
        wg_barrier();
        STAMP(2);
        double pap, rz, zz, rr0;
        double gz[NPL];                                       // a shard: z of this wave's slice on the ghost rows (from the rank neighbours)
        if constexpr (SHARD) {
            // Two-level meeting.  Every workgroup publishes its four sums to the workgroups of its rank (device memory); workgroup 0
            // of the rank adds them and stores the RANK's record into every rank's mailbox; every workgroup then adds the P rank
            // records in rank order — one local hop and one hop between GPUs per iteration (round 2: two meetings of P G records
            // each).  Meanwhile every wave takes the ghost rows of z of its slice from the own mailbox, the two boundary waves also
            // the neighbouring workgroup's boundary slice of z (own rows from that workgroup, ghost rows from the mailbox).
            const int rw = (W >= 3) ? 1 : 0;
            u64 *rankrec = Sh.mail[Sh.rank] + (size_t)SH_MAXREC * 2 + (size_t)par * 8 * REC4;      // [parity][P <= 8][8 granules] (the area of round 2's second meeting)
            bool ok = true;
            if (wv == rw) publish_rec4(slotsA, g, sum_part4(part, W, lane), epoch, lane);
            {   // ghost rows of the own slice (+ the halo slice of a boundary wave)
                const bool bw = G > 1 && (wv == 0 || wv == W - 1);
                const int th = (wv == 0) ? wrap(t0 - 1) : wrap(t0 + T);
                const u64 *bh = bw ? ((wv == 0) ? bnd + (((size_t)gm * 2 + 1) * HS) * 2 : bnd + (((size_t)gp * 2 + 0) * HS) * 2) : nullptr;
                const u64 *ga2[NPL];
#pragma unroll
                for (int q = 0; q < NPL; ++q) {
                    const int s = ss[q];
                    ga2[q] = !bw ? nullptr
                           : (s < Sh.own_lo) ? sh_ghost(Sh.mail[Sh.rank], 0, L, Sh.cap_ghost, th, s) + ghp
                           : (s >= Sh.own_hi && s < N) ? sh_ghost(Sh.mail[Sh.rank], 1, L, Sh.cap_ghost, th, s - Sh.own_hi) + ghp : nullptr;
                }
                u64 a0[NPL], a1[NPL], h0[NPL], h1[NPL], c0[NPL], c1[NPL];
                long long t_start = 0;
                for (int spin = 0;; ++spin) {
                    bool good = true;
#pragma unroll
                    for (int q = 0; q < NPL; ++q) {
                        if (gaddr[q]) { a0[q] = ld_mail<RANKS>(gaddr[q] + ghp); a1[q] = ld_mail<RANKS>(gaddr[q] + ghp + 1); }
                        if (bh) { h0[q] = ld_gran(bh + 2 * (lane + q * WAVE)); h1[q] = ld_gran(bh + 2 * (lane + q * WAVE) + 1); }
                        if (ga2[q]) { c0[q] = ld_mail<RANKS>(ga2[q]); c1[q] = ld_mail<RANKS>(ga2[q] + 1); }
                    }
#pragma unroll
                    for (int q = 0; q < NPL; ++q) {
                        if (gaddr[q]) good = good && (unsigned)(a0[q] >> 32) == epoch && (unsigned)(a1[q] >> 32) == epoch;
                        if (bh) good = good && (unsigned)(h0[q] >> 32) == epoch && (unsigned)(h1[q] >> 32) == epoch;
                        if (ga2[q]) good = good && (unsigned)(c0[q] >> 32) == epoch && (unsigned)(c1[q] >> 32) == epoch;
                    }
                    if (__all(good)) break;
                    if (poll_bail<NPL>(spin, t_start, lane, R)) { ok = false; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
#pragma unroll
                for (int q = 0; q < NPL; ++q) {
                    gz[q] = gaddr[q] ? __hiloint2double((int)(unsigned)a1[q], (int)(unsigned)a0[q]) : 0.0;
                    if (bw) {
                        const double hv = ga2[q] ? __hiloint2double((int)(unsigned)c1[q], (int)(unsigned)c0[q])
                                                 : __hiloint2double((int)(unsigned)h1[q], (int)(unsigned)h0[q]);
                        zhalo[((wv == 0) ? 0 : HSL) + lr + q * LSL] = hv;
                    }
                }
            }
            if (wv == rw) {
                if (g == 0) {       // the rank's record: the local records in workgroup order, then to every rank's mailbox
                    double d0, d1, t4 = 0.0;
                    if (G <= 8) { u64 v[1] = {0}; ok = poll_rec4<1>(slotsA, G, nullptr, nullptr, epoch, lane, R, v, d0, d1) && ok; t4 = sum_rec4<1>(v, G, lane); }
                    else        { u64 v[4] = {0, 0, 0, 0}; ok = poll_rec4<4>(slotsA, G, nullptr, nullptr, epoch, lane, R, v, d0, d1) && ok; t4 = sum_rec4<4>(v, G, lane); }
                    const double mine = __shfl(t4, lane & 6, WAVE);            // lane l: the total of value (l & 7) >> 1
                    if (lane < 8 * Sh.P) {
                        const u64 bits = (u64)__double_as_longlong(mine);
                        st_mail<RANKS>(Sh.mail[lane >> 3] + (size_t)SH_MAXREC * 2 + (size_t)par * 8 * REC4 + (size_t)Sh.rank * REC4 + (lane & 7),
                               ((u64)epoch << 32) | ((lane & 1) ? (bits >> 32) : (bits & 0xFFFFFFFFull)));
                    }
                }
                u64 v[1] = {0};
                long long t_start = 0;
                for (int spin = 0;; ++spin) {
                    bool good = true;
                    if (lane < REC4 * Sh.P) { v[0] = ld_mail<RANKS>(rankrec + lane); good = (unsigned)(v[0] >> 32) == epoch; }
                    if (__all(good)) break;
                    if (poll_bail<1>(spin, t_start, lane, R)) { ok = false; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
                const double t4 = sum_rec4<1>(v, Sh.P, lane);                  // ranks in rank order: the same bits on every rank
                if (lane < 8 && !(lane & 1)) tot[lane >> 1] = t4;
            }
            if (!ok && lane == 0) tot[5] = 0.0;
            wg_barrier();
            if (tot[5] == 0.0) return;
            pap = tot[0]; rz = tot[1]; zz = tot[2]; rr0 = tot[3];
        } else if (G == 1) {
            const double t4 = sum_part4(part, W, lane);
            pap = readlane_f64(t4, 0); rz = readlane_f64(t4, 8); zz = readlane_f64(t4, 16); rr0 = readlane_f64(t4, 24);
        } else {
            // Every wave fetches its share of the two boundary slices of z (2 NPL segments of 64 values, segment s -> wave s % W, two
            // at a time) into LDS; one of them also trades the team's records.
            constexpr int NSEG = 2 * NPL;
            const int rw = (W > NSEG) ? NSEG : ((W >= 3) ? 1 : 0);        // (a wave without a segment if there is one)
            bool ok = true;
            double t4 = 0.0;
#ifdef ELPH_WG_NOMEET
            // diagnostic bound (tools/time_wg_nomeet.py, never the product): an iteration WITHOUT its meeting — no record, no poll, the
            // workgroup's own sums taken for the team's, the boundary slices of z left as they are.  What any rearrangement of the
            // meeting (a pipelined recurrence, XCD-local records) could reach at most; the numbers it computes mean nothing.
            if (wv == rw) t4 = __shfl(sum_part4(part, W, lane), 8 * ((lane & 7) >> 1), WAVE);
            for (int s0 = wv; false;) {
#else
            if (wv == rw) publish_rec4(slotsA, g, sum_part4(part, W, lane), epoch, lane);
            for (int s0 = wv; s0 < NSEG || (s0 == wv && wv == rw); s0 += 2 * W) {
#endif
                const int s1 = s0 + W;
                const u64 *b0 = nullptr, *b1 = nullptr;
                if (s0 < NSEG) b0 = bnd + ((((s0 < NPL) ? (size_t)gm * 2 + 1 : (size_t)gp * 2 + 0) * HS) + lane_b + (size_t)(s0 % NPL) * WAVE) * 2;
                if (s1 < NSEG) b1 = bnd + ((((s1 < NPL) ? (size_t)gm * 2 + 1 : (size_t)gp * 2 + 0) * HS) + lane_b + (size_t)(s1 % NPL) * WAVE) * 2;
                const bool recs = (wv == rw && s0 == wv);
                double z0 = 0.0, z1 = 0.0;
                if (G <= 8) {
                    u64 v[1] = {0};
                    ok = poll_rec4<1>(recs ? slotsA : nullptr, G, b0, b1, epoch, lane, R, v, z0, z1) && ok;
                    if (recs) t4 = sum_rec4<1>(v, G, lane);
                } else {
                    u64 v[4] = {0, 0, 0, 0};
                    ok = poll_rec4<4>(recs ? slotsA : nullptr, G, b0, b1, epoch, lane, R, v, z0, z1) && ok;
                    if (recs) t4 = sum_rec4<4>(v, G, lane);
                }
                if (b0 && lwok) zhalo[(size_t)s0 * LSL + lr] = z0;        // (segment s = side * NPL + q lives at [side][q * LSL + slot])
                if (b1 && lwok) zhalo[(size_t)s1 * LSL + lr] = z1;
                if (!ok) break;
            }
            if (wv == rw && lane < 8 && !(lane & 1)) tot[lane >> 1] = t4;
            if (!ok && lane == 0) tot[5] = 0.0;
#ifdef ELPH_WG_ARRIVE
            if (!ok && lane == 0 && blockIdx.x < 4096) g_wg_arrive[blockIdx.x * 4 + 3] = ((unsigned long long)(seq + 1) << 8) | (1ull << wv);   // iteration and wave of a poll that gave up
#endif
            wg_barrier();
            if (tot[5] == 0.0) return;
            pap = tot[0]; rz = tot[1]; zz = tot[2]; rr0 = tot[3];
        }
        STAMP(1);
        rho = rr0;                                            // r.r of the residual this iteration started from, summed from the vector
#ifdef ELPH_WG_NOMEET
        const double alpha = 1e-6 * (rr0 / pap);              // (the workgroup's own sums are no CG: keep the vectors bounded — r almost constant, beta = 1/2 —
        rr = 0.5 * rr0 + 1e-30 * fabs(rr0 + alpha * (alpha * zz - 2.0 * rz));      //  so that no NaN sends a workgroup into the direct-sum meeting)
#else
        const double alpha = rr0 / pap;                                                            // :278-279 (rho = r.r)
        rr = rr0 + alpha * (alpha * zz - 2.0 * rz);
#endif
        // ---- x += alpha p, r -= alpha z (own slices) ----------------------------------------------------------------------------
#pragma unroll
        for (int j = 0; j < T; ++j)
#pragma unroll
            for (int q = 0; q < NPL; ++q) {
                const double zq = (SHARD && gaddr[q]) ? gz[q] : z[j][q];                           // (a shard's ghost rows: the owner's z)
                const double rn = rl[j * HSL + lr + q * LSL] - alpha * zq;                          // :285
                if (lwok) rl[j * HSL + lr + q * LSL] = rn;
                if (X_GLB) { if (lwok) xg[(size_t)(t0 + j) * N + sc[q]] = xr[X_REG ? j : 0][q] + alpha * p[j + 1][q]; }
                else if (X_REG) xr[X_REG ? j : 0][q] += alpha * p[j + 1][q];                       // :282
                else if (lwok) xl[j * HSL + lr + q * LSL] += alpha * p[j + 1][q];
            }
        if (G > 1 && (wv == 0 || wv == W - 1)) {              // the neighbouring workgroup's boundary slice of the new residual
            double *rh = rhalo + ((wv == 0) ? 0 : HSL);
            const double *zh = zhalo + ((wv == 0) ? 0 : HSL);
#pragma unroll
            for (int q = 0; q < NPL; ++q) if (lwok) rh[lr + q * LSL] = rh[lr + q * LSL] - alpha * zh[lr + q * LSL];
        }
        STAMP(3);
        if (!(rr > 1e-3 * rr0)) {
            // the identity cancels: take r'.r' from the vector itself (second meeting of this iteration; every wave of the team is here)
            double a = 0.0;
#pragma unroll
            for (int j = 0; j < T; ++j)
#pragma unroll
                for (int q = 0; q < NPL; ++q) if (own[q]) { const double rn = rl[j * HSL + lr + q * LSL]; a += rn * rn; }
            a = wave_sum_dpp(a);
            if (lane == 0) partF[wv] = a;
            wg_barrier();
            if constexpr (SHARD) {
                if (wv == 0) {
                    sh_publish<RANKS>(Sh, 0, g, G, wg_sum(partF, W, lane), epoch, lane);
                    const u64 *none[NPL];
#pragma unroll
                    for (int q = 0; q < NPL; ++q) none[q] = nullptr;
                    double t = 0.0, dummy[NPL];
                    const bool ok = sh_poll<NPL, RANKS>(Sh, true, 0, G, none, epoch, lane, R, t, dummy);
                    if (lane == 0) { tot[4] = t; if (!ok) tot[5] = 0.0; }
                }
                wg_barrier();
                if (tot[5] == 0.0) return;
                rr = tot[4];
            } else if (G == 1) {
                rr = wg_sum(partF, W, lane);
            } else {
                if (wv == 0) {
                    const double mine = wg_sum(partF, W, lane);
                    if (lane < 2) {
                        const u64 bits = (u64)__double_as_longlong(mine);
                        st_gran(slotsB + 2 * g + lane, ((u64)epoch << 32) | (lane ? (bits >> 32) : (bits & 0xFFFFFFFFull)));
                    }
                    u64 v = 0;
                    const bool ok = poll_records(slotsB, G, epoch, lane, R, v);
                    const int half = (int)(unsigned)v;
                    double t = 0.0;
                    for (int k = 0; k < G; ++k)
                        t += __hiloint2double(__builtin_amdgcn_readlane(half, 2 * k + 1), __builtin_amdgcn_readlane(half, 2 * k));
                    if (lane == 0) { tot[4] = t; if (!ok) tot[5] = 0.0; }
                }
                wg_barrier();
                if (tot[5] == 0.0) return;
                rr = tot[4];
            }
        }
        wg_barrier();                                         // the new residual of every wave is in LDS: the neighbours' halo slices
        STAMP(4);
        } else {
        // ================= two-meeting iteration (sharded solves) ===============================================================
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < T; ++j)
#pragma unroll
            for (int q = 0; q < NPL; ++q) if (own[q]) acc += p[j + 1][q] * z[j][q];
        acc = wave_sum_dpp(acc);
        // (x of the own slices lives in LDS)
        STAMP(0);
        // ---- meeting 1: p.z ---------------------------------------------------------------------------------------
        // wave partials -> LDS -> barrier; with a team of several workgroups ONE wave per workgroup publishes the workgroup's
        // record and polls the team's (80 waves polling one line serialise at the memory side: 1.5 -> ~0.7 us per meeting),
        // then hands the total to its workgroup through LDS
        if (lane == 0) partA[wv] = acc;
        wg_barrier();
        STAMP(2);
        double pap;
        if constexpr (SHARD) {
            if (wv == 0) {
                const double mine = wg_sum(partA, W, lane);
                sh_publish<RANKS>(Sh, 0, g, G, mine, epoch, lane);
                const u64 *none[NPL];
#pragma unroll
                for (int q = 0; q < NPL; ++q) none[q] = nullptr;
                double tot = 0.0, dummy[NPL];
                const bool ok = sh_poll<NPL, RANKS>(Sh, true, 0, G, none, epoch, lane, R, tot, dummy);
                if (lane == 0) { bc[0] = tot; if (!ok) bc[2] = 0.0; }
            }
            wg_barrier();
            if (bc[2] == 0.0) return;
            pap = bc[0];
        } else if (G == 1) {
            pap = wg_sum(partA, W, lane);
        } else {
            if (wv == 0) {
                const double mine = wg_sum(partA, W, lane);
                if (lane < 2) {
                    const u64 bits = (u64)__double_as_longlong(mine);
                    st_gran(slotsA + 2 * g + lane, ((u64)epoch << 32) | (lane ? (bits >> 32) : (bits & 0xFFFFFFFFull)));
                }
                u64 v = 0;
                const bool ok = poll_records(slotsA, G, epoch, lane, R, v);
                const int half = (int)(unsigned)v;
                double tot = 0.0;
                for (int k = 0; k < G; ++k)
                    tot += __hiloint2double(__builtin_amdgcn_readlane(half, 2 * k + 1), __builtin_amdgcn_readlane(half, 2 * k));
                if (lane == 0) { bc[0] = tot; if (!ok) bc[2] = 0.0; }
            }
            wg_barrier();
            if (bc[2] == 0.0) return;
            pap = bc[0];
        }
        STAMP(1);
        const double alpha = rho / pap;
        // ---- x += alpha p (to memory), r -= alpha z, r.r; show the boundary slices of the new r ---------------------------
        double a = 0.0, rn[T][NPL];
#pragma unroll
        for (int j = 0; j < T; ++j)
#pragma unroll
            for (int q = 0; q < NPL; ++q) {
                rn[j][q] = rl[j * HSL + lr + q * LSL] - alpha * z[j][q];                           // :285
                rl[j * HSL + lr + q * LSL] = rn[j][q];
                if (own[q]) a += rn[j][q] * rn[j][q];
                if (X_REG) xr[X_REG ? j : 0][q] += alpha * p[j + 1][q];                            // :282
                else xl[j * HSL + lr + q * LSL] += alpha * p[j + 1][q];
            }
        a = wave_sum_dpp(a);
        if constexpr (SHARD) {
            // the rows my neighbours hold as ghosts: straight into their mailboxes (device-initiated stores over xGMI)
            const int prev = (Sh.rank + Sh.P - 1) % Sh.P, next = (Sh.rank + 1) % Sh.P;
#pragma unroll
            for (int q = 0; q < NPL; ++q) {
                const int s = lane + q * WAVE;
                const u64 bits = (u64)__double_as_longlong(rn[0][q]), tag = (u64)epoch << 32;
                if (s >= Sh.own_lo && s < Sh.own_lo + Sh.n_to_prev) {              // bottom rows -> previous rank's ghosts above its own rows
                    u64 *d = sh_ghost(Sh.mail[prev], 1, L, Sh.cap_ghost, t0, s - Sh.own_lo);
                    st_mail<RANKS>(d, tag | (bits & 0xFFFFFFFFull)); st_mail<RANKS>(d + 1, tag | (bits >> 32));
                }
                if (s >= Sh.own_hi - Sh.n_to_next && s < Sh.own_hi) {              // top rows -> next rank's ghosts below its own rows
                    u64 *d = sh_ghost(Sh.mail[next], 0, L, Sh.cap_ghost, t0, s - (Sh.own_hi - Sh.n_to_next));
                    st_mail<RANKS>(d, tag | (bits & 0xFFFFFFFFull)); st_mail<RANKS>(d + 1, tag | (bits >> 32));
                }
            }
        }
        // slices that cross a workgroup boundary travel as granules too ({iteration, half of the f64}: the data is its own flag —
        // no drain here, no flag there; the neighbour polls them together with the r.r records)
        if (G > 1) {
            if (wv == 0) {
#pragma unroll
                for (int q = 0; q < NPL; ++q) st_f64_gran(bnd + (((size_t)g * 2 + 0) * HS + lane + q * WAVE) * 2, rn[0][q], epoch);
            }
            if (wv == W - 1) {
#pragma unroll
                for (int q = 0; q < NPL; ++q) st_f64_gran(bnd + (((size_t)g * 2 + 1) * HS + lane + q * WAVE) * 2, rn[T - 1][q], epoch);
            }
        }
        if (lane == 0) partB[wv] = a;
        STAMP(3);
        wg_barrier();
        STAMP(4);
        // ---- meeting 2: r.r and the halo slices of the new r ------------------------------------------------------------------
        if constexpr (SHARD) {
            // every wave takes the ghost rows of its slice from the mailbox; wave 0 also trades the r.r records of all ranks
            double tot = 0.0, gv[NPL];
            if (wv == 0) sh_publish<RANKS>(Sh, 1, g, G, wg_sum(partB, W, lane), epoch, lane);
            bool ok = sh_poll<NPL, RANKS>(Sh, wv == 0, 1, G, gaddr, epoch, lane, R, tot, gv);
#pragma unroll
            for (int q = 0; q < NPL; ++q) if (gaddr[q]) rl[lane + q * WAVE] = gv[q];
            if (G > 1 && (wv == 0 || wv == W - 1)) {          // the tau-neighbour workgroup of this rank: its boundary slice (own rows; ghosts follow below)
                u64 v = 0, gh[NPL][2];
                const u64 *bh = (wv == 0) ? bnd + (((size_t)gm * 2 + 1) * HS) * 2 : bnd + (((size_t)gp * 2 + 0) * HS) * 2;
                ok = poll_granules<NPL>(nullptr, G, bh, epoch, lane, R, v, gh) && ok;
#pragma unroll
                for (int q = 0; q < NPL; ++q) hx[q] = __hiloint2double((int)(unsigned)gh[q][1], (int)(unsigned)gh[q][0]);
                // ... whose ghost rows are as stale as mine were: the neighbour slice's ghost values come from the mailbox too
                const int th = (wv == 0) ? wrap(t0 - 1) : wrap(t0 + T);
                const u64 *ga2[NPL];
#pragma unroll
                for (int q = 0; q < NPL; ++q) {
                    const int s = lane + q * WAVE;
                    ga2[q] = (s < Sh.own_lo) ? sh_ghost(Sh.mail[Sh.rank], 0, L, Sh.cap_ghost, th, s)
                           : (s >= Sh.own_hi && s < N) ? sh_ghost(Sh.mail[Sh.rank], 1, L, Sh.cap_ghost, th, s - Sh.own_hi) : nullptr;
                }
                double t2 = 0.0, gv2[NPL];
                ok = sh_poll<NPL, RANKS>(Sh, false, 1, G, ga2, epoch, lane, R, t2, gv2) && ok;
#pragma unroll
                for (int q = 0; q < NPL; ++q) if (ga2[q]) hx[q] = gv2[q];
            }
            if (lane == 0) { if (wv == 0) bc[1] = tot; if (!ok) bc[2] = 0.0; }
            wg_barrier();
            if (bc[2] == 0.0) return;
            rr = bc[1];
        } else if (G == 1) {
            rr = wg_sum(partB, W, lane);
        } else {
            if (wv == 0 || wv == W - 1) {
                u64 v = 0, gh[NPL][2];
                const u64 *bh = (wv == 0) ? bnd + (((size_t)gm * 2 + 1) * HS) * 2                  // left halo: last slice of workgroup g - 1
                                          : bnd + (((size_t)gp * 2 + 0) * HS) * 2;                 // right halo: first slice of workgroup g + 1
                if (wv == 0) {
                    const double mine = wg_sum(partB, W, lane);
                    if (lane < 2) {
                        const u64 bits = (u64)__double_as_longlong(mine);
                        st_gran(slotsB + 2 * g + lane, ((u64)epoch << 32) | (lane ? (bits >> 32) : (bits & 0xFFFFFFFFull)));
                    }
                }
                const bool ok = poll_granules<NPL>(wv == 0 ? slotsB : nullptr, G, bh, epoch, lane, R, v, gh);
#pragma unroll
                for (int q = 0; q < NPL; ++q) hx[q] = __hiloint2double((int)(unsigned)gh[q][1], (int)(unsigned)gh[q][0]);
                if (wv == 0) {
                    const int half = (int)(unsigned)v;
                    double tot = 0.0;
                    for (int k = 0; k < G; ++k)
                        tot += __hiloint2double(__builtin_amdgcn_readlane(half, 2 * k + 1), __builtin_amdgcn_readlane(half, 2 * k));
                    if (lane == 0) bc[1] = tot;
                }
                if (!ok && lane == 0) bc[2] = 0.0;
            }
            wg_barrier();
            if (bc[2] == 0.0) return;
            rr = bc[1];
        }
        }   // two-meeting iteration
        STAMP(5);
        STAMP(6);
        // ---- stop test of iteration it = seq + 1 (IterativeSolvers.jl:286-295) ---------------------------------------------
        // eps = |r|/|b| < tol, kappa_min = max_j (2j / ln(2 eps0/eps_j))^2 > kappa_max, j = maxiter — a square root, two divisions and
        // a logarithm in f64 (~150 instructions in EVERY wave: half a mat-vec of this kernel).  Two comparisons screen them out
        // while no decision is near:  r.r well above (tol |b|)^2 rules out the first;  (2 eps0/eps)^2 = y outside [1/2, 2] means
        // |ln(2 eps0/eps)| > 0.34, which rules out the second while 2j < 0.34 sqrt(kappa_max).  Near a decision, on the last
        // iteration and with a residual history the exact arithmetic of the reference runs (eps, and kappa_min of THAT iteration:
        // an iteration stops on kappa only through its own term, the earlier ones did not stop).
        // (fixed_iters — measurement — takes the tolerance screen as passed: a solve would have stopped there; its iterations
        // cost what the iterations of a running solve cost.)
        const long long it = seq + 1;
        const bool fixed = R.fixed_iters > 0;
        int done = 0;
#ifdef ELPH_WG_NOMEET
        const bool screened = it < R.fixed_iters;             // (the sums mean nothing: no stop arithmetic before the last iteration)
#else
        const bool screened = !P.record_hist && it < (fixed ? R.fixed_iters : P.maxiter) && (fixed || rr > rr_far) &&
                              (rr + rr <= y_num || rr >= y_num + y_num) && (double)it < it_kappa;
#endif
        if (!screened) {
            eps = sqrt(rr) / normb;
            const double qq = 2.0 * (double)it / log(2.0 * eps0 / eps);
            const double val = qq * qq;
            kmin = (val > kmin) ? val : kmin;
            if (eps < P.tol) done = 1;
            else if (kmin > P.kmax) done = 2;
            else if (it >= P.maxiter) done = 3;
            if (fixed) done = (it >= R.fixed_iters) ? 3 : 0;
            if (g == 0 && wv == 0 && lane == 0 && P.record_hist) B.hist[(size_t)rhs * P.hist_stride + it] = eps;
        }
        STAMP(7);
        if (done) {
            STAMP_OUT(it);
            TL(2);
            // (the store addresses are made here, from a laundered lane number: computed before the loop they sit in 2 registers per
            //  value for the whole solve — in a kernel that has none to spare: 10 -> 2 spilled registers at 4 slices per wave)
            int lane2 = lane;
            asm volatile("" : "+v"(lane2));
#pragma unroll
            for (int j = 0; j < T; ++j)
#pragma unroll
                for (int q = 0; q < NPL; ++q) {
                    const int s2 = SQ ? sq_patch_site(lane2, q) : (HC ? hc_site(lane2, q) : (S8 ? s8_site(lane2) : (GR ? grid_site(lane2, q, GGX, GGY) : (HG ? hgrid_site<NPL>(lane2, q, HLX, HLY) : lane2 + q * WAVE))));
                    if (HC ? hc_real(lane2) : (GR ? lane2 < GGX * GGY : (HG ? lane2 < (HLX / HgDim<NPL>::PX) * (HLY / HgDim<NPL>::PY) : (SQ || s2 < N)))) {
                        // (the residual stays on the chip: ldiv! judges a solution by its TRUE residual, Models.jl:150-160; a shard's
                        //  caller may want it)
                        if (SHARD) rg[(size_t)(t0 + j) * N + s2] = rl[j * HSL + lr + q * LSL];
                        if (!X_GLB) xg[(size_t)(t0 + j) * N + s2] = X_REG ? xr[X_REG ? j : 0][q] : xl[j * HSL + lr + q * LSL];
                    }
                }
            if (g == 0 && wv == 0 && lane == 0) {
                CgState o = S;
                o.rho = rho; o.kmin = kmin; o.eps = eps; o.seq = it + 1; o.iters = it; o.done = done;
                st2[0] = o;
                st2[1] = o;
            }
            TL(3);
#ifdef ELPH_WG_PERSISTENT
            break;
#else
            return;
#endif
        }
        const double beta = rr / rho;
        rho = rr;
        // ---- next direction on the own slices and on the two halo slices (p = r + beta p is pointwise) --------------------
        {
            const bool lx = (G > 1 && wv == 0), rx = (G > 1 && wv == W - 1);
            const double *sl = rall + ((size_t)((wv > 0) ? wv - 1 : W - 1) * T + (T - 1)) * HSL;    // last slice of the wave below
            const double *sr = rall + ((size_t)((wv < W - 1) ? wv + 1 : 0) * T + 0) * HSL;         // first slice of the wave above
#pragma unroll
            for (int q = 0; q < NPL; ++q) {
                const double hl = lx ? (ONE ? rhalo[lr + q * LSL] : hx[q]) : sl[lr + q * LSL];
                const double hr = rx ? (ONE ? rhalo[HSL + lr + q * LSL] : hx[q]) : sr[lr + q * LSL];
                if constexpr (PH_LDS) {
                    const double n0 = hl + beta * PHALO(0, q), n1 = hr + beta * PHALO(1, q);
                    if (lwok) { PHALO(0, q) = n0; PHALO(1, q) = n1; }
                } else {
                    p[0][q] = hl + beta * p[0][q];
                    p[T + 1][q] = hr + beta * p[T + 1][q];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < T; ++j)
#pragma unroll
            for (int q = 0; q < NPL; ++q) p[j + 1][q] = rl[j * HSL + lr + q * LSL] + beta * p[j + 1][q];
        STAMP(8);
    }
#ifdef ELPH_WG_PERSISTENT
    if (SHARD) return;
    wg_barrier();                                      // (the LDS of this right-hand side is rewritten by the next one)
#endif
    }
#undef EXPV
#undef PHALO
}

// ------------------------------------------------------------------------------------------------------------------------
// THE ROW FORM (round 5): Holstein with uniform hopping on the 16 x 16 square lattice (config C), batches that take the 4-slices-per-wave
// shape.  Same team protocol as k_cg_wg's single-meeting iteration (one 64-byte record per workgroup, boundary slices of z as granules,
// r'.r' by the one-step identity, the same stop test) around ANOTHER layout of the vectors: a 16-lane DPP ROW holds ONE time slice —
// lane (X, Y) = (l & 3, (l >> 2) & 3) a 4 x 4 patch of its sites, register q = cx + 4 cy — and the four rows of a wave four consecutive
// slices, 32 slices per workgroup.  Per colour the even bonds and the inner odd bonds pair registers of a lane; only the patch edges
// cross: x-odd by quad permutations, y-odd by row rotations — 16 DPP moves of an f64 per sweep of a slice, all inside the row (the 2 x 2
// patches of k_cg_wg: 12 moves + 4 ds_bpermute per SLAB sweep, nine slab sweeps per mat-vec pair of four slices).  The tau shift, which
// there is a renaming of registers (at the price of one halo sweep per wave), goes through LDS here: p of the slice below, then
// t = sg k4 E o CB^T(M p) of the slice above, 16 ds_write_b64 + 16 ds_read_b64 each and a barrier.  Only the workgroup's TOP boundary needs
// a redundant slice (m and t of the slice above its last one): one extra sweep pair, by the last wave.
// Measured before it was built (tools/probes/matvec_row_layout_probe.cpp): 2.4 us per mat-vec pair + sums + updates of a round of the
// chip against ~3.9 us of the 2 x 2 layout.
namespace rowf {
constexpr int SV = 256;                       // values of a slice
template <int CTRL>
__device__ __forceinline__ double dpp(double v) { return dpp_f64<CTRL>(v); }
// one colour of the checkerboard on the 4 x 4 patch v[cx + 4 cy]: v <- (I + th P_colour) v.  COL 0: x-even, 1: x-odd, 2: y-even, 3: y-odd
template <int COL>
__device__ __forceinline__ void colour(double (&v)[16], double th) {
    if constexpr (COL == 0) {
#pragma unroll
        for (int cy = 0; cy < 4; ++cy)
#pragma unroll
            for (int cx = 0; cx < 4; cx += 2) { const int i = cx + 4 * cy, j = i + 1; const double a = v[i] + th * v[j], b = v[j] + th * v[i]; v[i] = a; v[j] = b; }
    } else if constexpr (COL == 2) {
#pragma unroll
        for (int cx = 0; cx < 4; ++cx)
#pragma unroll
            for (int cy = 0; cy < 4; cy += 2) { const int i = cx + 4 * cy, j = i + 4; const double a = v[i] + th * v[j], b = v[j] + th * v[i]; v[i] = a; v[j] = b; }
    } else if constexpr (COL == 1) {
        double fu[4], fd[4];
#pragma unroll
        for (int cy = 0; cy < 4; ++cy) { fu[cy] = dpp<0x39>(v[0 + 4 * cy]); fd[cy] = dpp<0x93>(v[3 + 4 * cy]); }      // quad_perm [1,2,3,0]: from the patch X + 1; [3,0,1,2]: from X - 1
#pragma unroll
        for (int cy = 0; cy < 4; ++cy) { const int i = 1 + 4 * cy, j = i + 1; const double a = v[i] + th * v[j], b = v[j] + th * v[i]; v[i] = a; v[j] = b; }
#pragma unroll
        for (int cy = 0; cy < 4; ++cy) { v[3 + 4 * cy] += th * fu[cy]; v[0 + 4 * cy] += th * fd[cy]; }
    } else {
        double fu[4], fd[4];
#pragma unroll
        for (int cx = 0; cx < 4; ++cx) { fu[cx] = dpp<0x12C>(v[cx + 0]); fd[cx] = dpp<0x124>(v[cx + 12]); }            // row_ror:12: from lane + 4 (the patch Y + 1); row_ror:4: from lane - 4
#pragma unroll
        for (int cx = 0; cx < 4; ++cx) { const int i = cx + 4, j = i + 4; const double a = v[i] + th * v[j], b = v[j] + th * v[i]; v[i] = a; v[j] = b; }
#pragma unroll
        for (int cx = 0; cx < 4; ++cx) { v[cx + 12] += th * fu[cx]; v[cx + 0] += th * fd[cx]; }
    }
    __builtin_amdgcn_sched_barrier(0);
}
template <bool REVERSE>
__device__ __forceinline__ void sweep(double (&v)[16], double th) {
    if constexpr (!REVERSE) { colour<0>(v, th); colour<1>(v, th); colour<2>(v, th); colour<3>(v, th); }
    else { colour<3>(v, th); colour<2>(v, th); colour<1>(v, th); colour<0>(v, th); }
}
}  // namespace rowf

template <bool X0Z>
__global__ void __launch_bounds__(512) k_cg_row(CgBufs B, ModelDev m, WgCtl R) {
    using rowf::SV;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int W = 8, NPLM = 4, HS = 256, LSL = WAVE;      // (the meeting sees a slice as four segments of 64 values, as k_cg_wg's 2 x 2 layout does)
    const int G = R.G;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int tq = idx / G, g = idx - tq * G;
    const int rhs = tq * 8 + xcd;
    if (rhs >= B.nrhs) return;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & (WAVE - 1), row = lane >> 4, l16 = lane & 15;
    const int N = m.N, L = m.L;
    const int vr = 4 * wv + row;                       // this lane's row of the workgroup: 0 .. 31
    const int t = g * 32 + vr;                         // ... and its time slice
    const size_t ndim = (size_t)N * L;
    // LDS: XB[34][256] exchange rows (row 0: p of the slice below the workgroup, kept; 1..32: the rows' p, then t; 33: t of the slice above) |
    //      RL[34][256] r (0 and 33: the neighbouring workgroups' boundary slices) | PH1[256] p of the slice above | EH[256] its exp(-dtau V) |
    //      zhalo[2][256] | part[32] tot[8] partF[8]
    double *XB = lds, *RL = XB + 34 * SV, *PH1 = RL + 34 * SV, *EH = PH1 + SV, *zhalo = EH + SV, *part = zhalo + 2 * SV, *tot = part + 32, *partF = tot + 8;
    auto wrap = [L](int tt) { return (tt < 0) ? tt + L : ((tt >= L) ? tt - L : tt); };
    auto sgn = [](int tt) { return (tt == 0) ? -1.0 : 1.0; };
    const CgParams P = B.params;
    CgState *st2 = B.state + 2 * rhs;
    const CgState S = ld_state(st2);
    if (S.done || S.seq != 0) return;                  // fresh solves only (the host guarantees it)
    // site of register q of a lane: (4 X + cx) + 16 (4 Y + cy) — made where it is used (set-up, final store), not kept across the loop
    auto site_of = [](int l16v, int q) { return (4 * (l16v & 3) + (q & 3)) + 16 * (4 * (l16v >> 2) + (q >> 2)); };
    double *xg = B.x + (size_t)rhs * ndim;
    const double *rg = B.r + (size_t)rhs * ndim;
    const double *Ech = m.E + (size_t)(rhs % m.nchains) * m.E_chain_stride;
    const double th = m.s_uni / m.c_uni, k4 = (m.c_uni * m.c_uni) * (m.c_uni * m.c_uni);
    double p[16], x[16], e[16], z[16];      // (x in memory instead — loaded behind the sums, stored by the update — was measured slower: 9.3 against 8.0 us)
    const int own = (1 + vr) * SV + l16;               // this lane's words of its row: own + 16 q
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const double rv = rg[(size_t)t * N + site_of(l16, q)];
        RL[own + 16 * q] = rv;
        p[q] = rv;                                     // p0 = r0 (IterativeSolvers.jl:272)
        x[q] = X0Z ? 0.0 : xg[(size_t)t * N + site_of(l16, q)];
        e[q] = Ech[(size_t)t * m.E_tau_stride + site_of(l16, q)];
    }
    if (vr == 0) {                                     // the slice below the workgroup: p0 = r0 of it
        const int tb = wrap(g * 32 - 1);
#pragma unroll
        for (int q = 0; q < 16; ++q) { const double rv = rg[(size_t)tb * N + site_of(l16, q)]; XB[l16 + 16 * q] = rv; RL[l16 + 16 * q] = rv; }
    }
    if (vr == 31) {                                    // the slice above
        const int ta = wrap(g * 32 + 32);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const double rv = rg[(size_t)ta * N + site_of(l16, q)];
            PH1[l16 + 16 * q] = rv; RL[33 * SV + l16 + 16 * q] = rv;
            EH[l16 + 16 * q] = Ech[(size_t)ta * m.E_tau_stride + site_of(l16, q)];
        }
    }
    if (threadIdx.x == 0) tot[5] = 1.0;
    u64 *const slots0 = R.slots + (size_t)rhs * 2 * SLOTS_PER_RHS;
    u64 *const bnd0 = R.bnd + (size_t)rhs * 2 * G * 2 * HS * 2;     // [parity][G][first | last slice][HS][2 granules]
    const int gm = (g == 0) ? G - 1 : g - 1, gp = (g == G - 1) ? 0 : g + 1;
    double rho = S.rho, kmin = S.kmin, eps = S.eps;
    const double eps0 = S.eps0, normb = S.normb;
    const double rr_far = (P.tol * normb) * (P.tol * normb) * 1.000001, y_num = 4.0 * (eps0 * normb) * (eps0 * normb);
    const double it_kappa = 0.17 * sqrt(P.kmax);
    const double sg_own = sgn(t) * k4, sg_top = sgn(wrap(g * 32 + 32)) * k4;
    wg_barrier();
    STAMP_DECL;
    for (long long seq = 0;; ++seq) {
        const unsigned epoch = R.epoch0 + (unsigned)seq + 1u;
        const unsigned par = epoch & 1u;
        u64 *const slotsA = slots0 + (size_t)par * SLOTS_PER_RHS, *const slotsB = slotsA + SLOTS_A;
        u64 *const bnd = bnd0 + (size_t)par * G * 2 * HS * 2;
        STAMP(9);
        // ---- z = M^T M p:  m(t) = p(t) - sg(t) k4 CB [E(t) o p(t-1)];  tv(t) = sg(t) k4 E(t) o CB^T m(t);  z(t) = m(t) - tv(t+1)
#pragma unroll
        for (int q = 0; q < 16; ++q) XB[own + 16 * q] = p[q];
        wg_barrier();
        double u[16];
#ifndef ELPH_ROW_NOHALO      // (timing experiment, wrong results: without the redundant slice of the last wave)
        if (wv == W - 1) {
            // the slice ABOVE the workgroup first (every row of this wave computes it, identically; in the registers the own slice uses next): its
            // m from the kept p of that slice and the p of the workgroup's last slice (in its exchange row), then its tv into exchange row 33
#pragma unroll
            for (int q = 0; q < 16; ++q) u[q] = EH[l16 + 16 * q] * XB[32 * SV + l16 + 16 * q];
            rowf::sweep<false>(u, th);
#pragma unroll
            for (int q = 0; q < 16; ++q) u[q] = PH1[l16 + 16 * q] - sg_top * u[q];       // m of the slice above
            rowf::sweep<true>(u, th);
#pragma unroll
            for (int q = 0; q < 16; ++q) XB[33 * SV + l16 + 16 * q] = sg_top * (EH[l16 + 16 * q] * u[q]);
        }
#endif
#pragma unroll
        for (int q = 0; q < 16; ++q) u[q] = e[q] * XB[own - SV + 16 * q];            // p of the slice below (row 0: the kept halo)
        rowf::sweep<false>(u, th);
#pragma unroll
        for (int q = 0; q < 16; ++q) { z[q] = p[q] - sg_own * u[q]; u[q] = z[q]; }   // z holds m for now
        rowf::sweep<true>(u, th);
#pragma unroll
        for (int q = 0; q < 16; ++q) u[q] = sg_own * (e[q] * u[q]);                  // tv of the own slice
        wg_barrier();                                  // everybody has read the p of the row below: the rows take the tv's now
#pragma unroll
        for (int q = 0; q < 16; ++q) XB[own + 16 * q] = u[q];
        wg_barrier();
        double s_pz = 0.0, s_rz = 0.0, s_zz = 0.0, s_rr = 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            z[q] = z[q] - XB[own + SV + 16 * q];       // z(t) = m(t) - tv(t+1)
            const double rv = RL[own + 16 * q];
            s_pz += p[q] * z[q]; s_rz += rv * z[q]; s_zz += z[q] * z[q]; s_rr += rv * rv;
        }
        {
            const double k4s = wave_sum4(s_pz, s_rz, s_zz, s_rr, lane);
            if (lane < 4) part[lane * 8 + wv] = k4s;
        }
        STAMP(0);
        // boundary slices of z for the neighbouring workgroups (self-tagged granules; word i of a slice = 16 q + l16, as both sides see it)
        if (vr == 0) {
#pragma unroll
            for (int q = 0; q < 16; ++q) st_f64_gran(bnd + (((size_t)g * 2 + 0) * HS + l16 + 16 * q) * 2, z[q], epoch);
        }
        if (vr == 31) {
#pragma unroll
            for (int q = 0; q < 16; ++q) st_f64_gran(bnd + (((size_t)g * 2 + 1) * HS + l16 + 16 * q) * 2, z[q], epoch);
        }
        wg_barrier();
        STAMP(2);
        double pap, rz, zz, rr0;
        {
            // the meeting, as k_cg_wg's: wave s takes segment s of the two boundary slices (8 segments of 64 values), wave 1 the records too
            constexpr int NSEG = 2 * NPLM, rw = 1;
            if (wv == rw) publish_rec4(slotsA, g, sum_part4(part, W, lane), epoch, lane);
            bool ok = true;
            double t4 = 0.0;
            {
                const int s0 = wv;
                const u64 *b0 = bnd + ((((s0 < NPLM) ? (size_t)gm * 2 + 1 : (size_t)gp * 2 + 0) * HS) + lane + (size_t)(s0 % NPLM) * WAVE) * 2;
                const bool recs = (wv == rw);
                double z0 = 0.0, z1 = 0.0;
                if (G <= 8) {
                    u64 v[1] = {0};
                    ok = poll_rec4<1>(recs ? slotsA : nullptr, G, b0, nullptr, epoch, lane, R, v, z0, z1) && ok;
                    if (recs) t4 = sum_rec4<1>(v, G, lane);
                } else {
                    u64 v[4] = {0, 0, 0, 0};
                    ok = poll_rec4<4>(recs ? slotsA : nullptr, G, b0, nullptr, epoch, lane, R, v, z0, z1) && ok;
                    if (recs) t4 = sum_rec4<4>(v, G, lane);
                }
                zhalo[(size_t)s0 * LSL + lane] = z0;   // (segment s = side * 4 + k lives at [side][64 k + lane])
                static_assert(NSEG == W, "one segment per wave");
            }
            if (wv == rw && lane < 8 && !(lane & 1)) tot[lane >> 1] = t4;
            if (!ok && lane == 0) tot[5] = 0.0;
            wg_barrier();
            if (tot[5] == 0.0) return;
            pap = tot[0]; rz = tot[1]; zz = tot[2]; rr0 = tot[3];
        }
        STAMP(1);
        rho = rr0;
        const double alpha = rr0 / pap;
        double rr = rr0 + alpha * (alpha * zz - 2.0 * rz);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            RL[own + 16 * q] = RL[own + 16 * q] - alpha * z[q];
            x[q] += alpha * p[q];
        }
        if (wv == 0 || wv == W - 1) {                  // the neighbouring workgroups' boundary slices of the new residual
            double *rh = RL + ((wv == 0) ? 0 : 33 * SV);
            const double *zh = zhalo + ((wv == 0) ? 0 : SV);
#pragma unroll
            for (int k = 0; k < 4; ++k) rh[lane + 64 * k] = rh[lane + 64 * k] - alpha * zh[lane + 64 * k];
        }
        if (!(rr > 1e-3 * rr0)) {
            // the identity cancels: r'.r' from the vector itself (second meeting of this iteration; every wave of the team is here)
            double a = 0.0;
#pragma unroll
            for (int q = 0; q < 16; ++q) { const double rn = RL[own + 16 * q]; a += rn * rn; }
            a = wave_sum_dpp(a);
            if (lane == 0) partF[wv] = a;
            wg_barrier();
            if (wv == 0) {
                const double mine = wg_sum(partF, W, lane);
                if (lane < 2) {
                    const u64 bits = (u64)__double_as_longlong(mine);
                    st_gran(slotsB + 2 * g + lane, ((u64)epoch << 32) | (lane ? (bits >> 32) : (bits & 0xFFFFFFFFull)));
                }
                u64 v = 0;
                const bool ok = poll_records(slotsB, G, epoch, lane, R, v);
                const int half = (int)(unsigned)v;
                double tt = 0.0;
                for (int k = 0; k < G; ++k)
                    tt += __hiloint2double(__builtin_amdgcn_readlane(half, 2 * k + 1), __builtin_amdgcn_readlane(half, 2 * k));
                if (lane == 0) { tot[4] = tt; if (!ok) tot[5] = 0.0; }
            }
            wg_barrier();
            if (tot[5] == 0.0) return;
            rr = tot[4];
        }
        STAMP(3);
        // ---- stop test of iteration it = seq + 1 (IterativeSolvers.jl:286-295), screened as in k_cg_wg
        const long long it = seq + 1;
        const bool fixed = R.fixed_iters > 0;
        int done = 0;
        const bool screened = !P.record_hist && it < (fixed ? R.fixed_iters : P.maxiter) && (fixed || rr > rr_far) &&
                              (rr + rr <= y_num || rr >= y_num + y_num) && (double)it < it_kappa;
        if (!screened) {
            eps = sqrt(rr) / normb;
            const double qq = 2.0 * (double)it / log(2.0 * eps0 / eps);
            const double val = qq * qq;
            kmin = (val > kmin) ? val : kmin;
            if (eps < P.tol) done = 1;
            else if (kmin > P.kmax) done = 2;
            else if (it >= P.maxiter) done = 3;
            if (fixed) done = (it >= R.fixed_iters) ? 3 : 0;
            if (g == 0 && wv == 0 && lane == 0 && P.record_hist) B.hist[(size_t)rhs * P.hist_stride + it] = eps;
        }
        STAMP(7);
        if (done) {
            STAMP_OUT(it);
            int l16b = l16;
            asm volatile("" : "+v"(l16b));             // (addresses made here, from a laundered lane number: not 32 registers across the loop)
#pragma unroll
            for (int q = 0; q < 16; ++q) xg[(size_t)t * N + site_of(l16b, q)] = x[q];
            if (g == 0 && wv == 0 && lane == 0) {
                CgState o = S;
                o.rho = rho; o.kmin = kmin; o.eps = eps; o.seq = it + 1; o.iters = it; o.done = done;
                st2[0] = o;
                st2[1] = o;
            }
            return;
        }
        const double beta = rr / rho;
        rho = rr;
        // ---- next direction: own slice, and the two kept slices of the neighbouring workgroups (p = r + beta p is pointwise)
#pragma unroll
        for (int q = 0; q < 16; ++q) p[q] = RL[own + 16 * q] + beta * p[q];
        if (wv == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) XB[lane + 64 * k] = RL[lane + 64 * k] + beta * XB[lane + 64 * k];
        }
        if (wv == W - 1) {
#pragma unroll
            for (int k = 0; k < 4; ++k) PH1[lane + 64 * k] = RL[33 * SV + lane + 64 * k] + beta * PH1[lane + 64 * k];
        }
        STAMP(8);
    }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------

struct Shape { int T, W, G; size_t shm; bool sq, hc, s8, gr, hg; int npl; bool tg = false; };   // npl: sites per lane of the kernel (honeycomb DPP form: 6; grid form: 4)

// DPP form: Holstein on the 16 x 16 square lattice in the reference's colouring (detect_square)
static bool sq_form(const elph_handle_s *h, const ModelDev &m) {
    const char *e = getenv("ELPH_WG_NO_DPP");
    return h->sq_P == 2 && h->N == 256 && m.sq_bond && !(e && e[0] == '1');
}

// T slices per wave: what the register file takes at two waves per SIMD — lane-program form 2 for site phonons with <= 4 sites
// per lane, else 1; DPP form 2 (config C: 5.8 us per iteration, 24 right-hand sides fill the chip), or 4 for batches that the
// 2-slice shape cannot hold in one round (6.8 us per iteration, but 48 right-hand sides per round: 14 vs 7.3 M mat-vecs/s);
// W = the largest divisor of Ltau / T that is <= 8 waves; G = workgroups per right-hand side (<= 32: the 2G record granules of
// a meeting are polled by one wave instruction)
// honeycomb DPP form: Holstein with uniform hopping on 12 x 12 cells in the reference's colouring (detect_honeycomb12)
static bool hc_form(const elph_handle_s *h, const ModelDev &m) {
    const char *e = getenv("ELPH_WG_NO_DPP");
    return h->kind == ELPH_MODEL_HOLSTEIN && h->hc12 && m.uniform && !(e && e[0] == '1');
}

// 8 x 8 DPP form: Holstein with uniform hopping on the 8 x 8 square lattice in the reference's colouring (detect_square: sq_P = 1)
static bool s8_form(const elph_handle_s *h, const ModelDev &m) {
    const char *e = getenv("ELPH_WG_NO_DPP");
    return h->kind == ELPH_MODEL_HOLSTEIN && h->sq_P == 1 && h->N == 64 && m.uniform && !(e && e[0] == '1');
}

// grid form: Holstein with uniform hopping on a periodic LX x LY square lattice in the reference's colouring whose 2 x 2 patches fit the
// lanes of a wave (detect_square: sq_LX, sq_LY).  An ordinary solve takes it for the sizes WITHOUT a DPP form of their own (not 16 x 16,
// not 8 x 8); a sharded solve (for_shard) for every size — the slab closed into a ring is such a lattice, and the DPP forms know no shards.
static bool gr_form(const elph_handle_s *h, const ModelDev &m, bool for_shard = false) {
    const char *e = getenv("ELPH_WG_NO_DPP");
    if (h->kind != ELPH_MODEL_HOLSTEIN || h->sq_LX < 4 || h->sq_LY < 4 || !m.uniform || m.grid_GX * m.grid_GY < 1 || m.grid_GX * m.grid_GY > 64 || (e && e[0] == '1')) return false;
    return for_shard || h->sq_P == 0;
}

// triangular grid form: Holstein with uniform hopping on an even-L triangular lattice of at most 16 x 16 sites in the reference's colouring
// (detect_triangular: pg_kind 3): the GRID layout with the two diagonal colours (pgrid::Tri<2, 2>)
static bool tg_form(const elph_handle_s *h, const ModelDev &m) {
    const char *e = getenv("ELPH_WG_NO_DPP");
    return h->kind == ELPH_MODEL_HOLSTEIN && h->pg_kind == 3 && h->pg_L >= 4 && h->pg_L <= 16 && m.uniform && m.grid_GX == h->pg_L / 2 && m.grid_GY == h->pg_L / 2 && !(e && e[0] == '1');
}

// honeycomb grid form: Holstein with uniform hopping on a periodic honeycomb lattice of LX x LY cells in the reference's colouring whose cells
// fit a grid of lanes (detect_honeycomb: hc_LX, hc_LY; 12 x 12 has a DPP form of its own — which knows no shards).  Returns the registers per
// lane (2, 4 or 8: 1, 2 x 1 or 2 x 2 cells), 0: no.
static int hg_form(const elph_handle_s *h, const ModelDev &m, bool for_shard = false) {
    const char *e = getenv("ELPH_WG_NO_DPP");
    if (h->kind != ELPH_MODEL_HOLSTEIN || h->hc_LX < 2 || h->hc_LY < 2 || (h->hc12 && !for_shard) || !m.uniform || m.hc_LX != h->hc_LX || (e && e[0] == '1')) return 0;
    const int LX = h->hc_LX, LY = h->hc_LY;
    if (LX * LY <= 64) return 2;
    if (LX % 2 == 0 && (LX / 2) * LY <= 64) return 4;
    if (LX % 2 == 0 && LY % 2 == 0 && (LX / 2) * (LY / 2) <= 64) return 8;
    return 0;
}

static int largest_divisor_le8(int n, int cap = 8) { for (int w = std::min(cap, n); w >= 1; --w) if (n % w == 0) return w; return 1; }

static bool pick_shape(const elph_handle_s *h, const ModelDev &m, int forceT, int nrhs, Shape *out) {
    const int L = (int)h->L;
    const bool ssh = (h->kind == ELPH_MODEL_SSH), sq = sq_form(h, m), hc = !sq && hc_form(h, m), s8 = !sq && !hc && s8_form(h, m);
    const bool tg = !sq && !hc && !s8 && tg_form(h, m);
    const bool gr = !sq && !hc && !s8 && (tg || gr_form(h, m));
    const int hgn = (!sq && !hc && !s8 && !gr) ? hg_form(h, m) : 0;
    const bool hg = hgn > 0;
    const int npl = hc ? HC_NPL : (gr ? 4 : (hg ? hgn : h->npl));
    if (h->lp_mc != 4 && !tg) return false;               // (a six-colour lattice has no lane-program form here: only the triangular grid form)
    if (!hc && !gr && !hg && h->npl > 5) return false;     // (the lane-program form carries at most 320 sites; elph_wg_usable lets larger honeycomb lattices through for the grid form only)
    // one site per lane (the 8 x 8 lattice: config B): the whole time axis fits ONE workgroup — up to 8 waves of 4, 5 or 8 slices — and
    // a team of one needs no records, no boundary granules, no polls: its meeting is an LDS reduction and a barrier, and a round holds
    // 256 right-hand sides.  Its iteration is longer (config B: 6.3 us at 5 slices per wave against 3.5 at 2 with teams of four), so it
    // is the shape of batches beyond one round of 2 slices per wave (B: 64): 256 right-hand sides 14.0 -> 6.4 us = 36 -> 80 M mat-vecs/s
    bool big_batch = true;
    if (!forceT && L % 2 == 0) {
        const int G2 = (L / 2) / largest_divisor_le8(L / 2);
        big_batch = G2 <= 32 && nrhs > 8 * (32 / G2);
    }
    // (the 8 x 8 DPP form: the team of one is the fastest shape at every batch size — 2.04 us per iteration for one right-hand side, 2.3 us
    //  for 256 = 222 M mat-vecs/s — its sweeps are a few dozen register moves)
    if (!ssh && !sq && !hc && !gr && !hg && h->npl == 1 && (big_batch || s8)) {
        const int one[3] = {4, 5, 8};
        for (int T : one) {
            if ((forceT && T != forceT) || L % T || L / T > 8 || L / T < 2) continue;
            const int W = L / T;
            const size_t SL = (size_t)npl * WAVE + 2 * WAVE, HS = (size_t)npl * WAVE;
            const size_t shm = ((size_t)W * (s8 ? 0 : T + 1) * SL + 2 * (size_t)W * T * HS + 48 + 4 * HS) * sizeof(double);
            if (shm > 160 * 1024) continue;
            out->T = T; out->W = W; out->G = 1; out->shm = shm; out->sq = false; out->hc = false; out->s8 = s8; out->gr = false; out->hg = false; out->npl = npl;
            return true;
        }
    }
    const int cand[4] = {4, 3, 2, 1};
    for (int T : cand) {
        if (forceT && T != forceT) continue;
        if (T == 3) {
            // honeycomb form only: 48 right-hand sides per round instead of 24 at 2 slices per wave, but 8.6 us per iteration against 5.7
            // (the kernel still spills 42 registers: six sites per lane x three slices is more than the register file and the LDS hold
            // next to each other, even with x in memory) — taken when the rounds it saves outweigh that: 25-48 and everything from 73
            // right-hand sides on
            if (!hc || L % 3 || L % 2) continue;
            if (forceT != 3) {
                const int G2 = (L / 2) / largest_divisor_le8(L / 2), G3 = (L / 3) / largest_divisor_le8(L / 3);
                if (G2 > 32 || G3 > 32) continue;
                const int r2 = 8 * (32 / G2), r3 = 8 * (32 / G3);
                if (nrhs <= r2 || 8.6 * ((nrhs + r3 - 1) / r3) >= 5.7 * ((nrhs + r2 - 1) / r2)) continue;
            }
        }
        if (T == 4) {
            if (!(sq || gr) || !m.uniform || ssh) continue;
            if (forceT != 4) {                           // only when 2 slices per wave would need a second round: 8 XCDs x (32 CUs / G2) teams
                if (L % 2) continue;
                const int G2 = (L / 2) / largest_divisor_le8(L / 2);
                if (G2 > 32 || nrhs <= 8 * (32 / G2)) continue;
            }
        }
        if (gr && T == 3) continue;
        if (hg && T > 2) continue;
        if (T == 2 && (sq || hc || gr || hg) && forceT != 2) {
            // DPP form: a batch that one round of 1 slice per wave holds (config C: up to 8 right-hand sides, one team of 20 workgroups per
            // XCD) runs that shape — with ONE meeting per iteration the shorter mat-vec wins over the larger team: 3.47 against 4.00 us
            // per iteration for one right-hand side (round 2, two meetings: the other way round)
            const int G1 = L / largest_divisor_le8(L);
            if (G1 <= 32 && L / G1 >= 2 && nrhs <= 8 * (32 / G1)) continue;
        }
        if (T == 2 && !sq && !hc && !gr && !hg && ((ssh && h->npl > 4) || h->npl > 5 || (!ssh && h->npl >= 4 && !m.uniform))) continue;
        if (T == 2 && !sq && !hc && !gr && !hg && (h->npl == 5 || ssh) && forceT != 2) {
            // 5 sites per lane (honeycomb L = 12) and bond phonons (three table sets per wave: 41 registers spill): 2 slices per wave are
            // slower per iteration (D: 10.0 vs 9.3 us; E: 12.9 vs 9.5 us) but hold more right-hand sides per round (D: 24 instead
            // of 16; E: 24 instead of 8) — taken once a batch exceeds the round of 1 slice per wave
            const int G1 = L / largest_divisor_le8(L);
            if (G1 <= 32 && nrhs <= 8 * (32 / G1)) continue;
        }
        if (L % T) continue;
        const int Wt = L / T;
        int W = 0;
        const char *ew = getenv("ELPH_WG_W");              // (experiments: cap on the waves per workgroup)
        const int wmax = ew ? std::max(1, atoi(ew)) : 8;
        for (int w = std::min(wmax, Wt); w >= 1; --w) if (Wt % w == 0) { W = w; break; }
        const int G = Wt / W;
        if (G > 32) continue;
        if (G > 1 && W < 2) continue;                    // (a wave polls at most ONE neighbouring workgroup's boundary slice)
        const size_t SL = (size_t)npl * WAVE + 2 * WAVE, HS = (size_t)npl * (hc ? 48 : WAVE);     // (the kernel's HSL: a slice in LDS)
        const bool s_lds = sq && ssh && T == 2;            // (the kernel's S_LDS: table set of slice t0 in LDS, x in registers)
        const bool e_lds = ((sq || gr) && T >= 4) || (hc && T >= 2), x_reg = ((sq || gr) && T >= 4) || s_lds || (hc && T >= 3);
        const size_t shm = ((size_t)W * ((sq || hc || s8 || gr || hg) ? 0 : T + 1) * SL + (size_t)(x_reg ? 1 : 2) * W * T * HS +
                            (e_lds ? (hc ? ((size_t)W * T + 1) * HS : (size_t)W * (T + 1) * HS) : 0) + (s_lds ? (size_t)W * wg::SQ_TABS * WAVE : 0) +
                            48 + 4 * HS + ((hc && T >= 3) ? (size_t)W * 2 * HS : 0)) * sizeof(double);   // + partials, totals, rhalo[2][HS], zhalo[2][HS] (+ the halo slices of p)
        if (shm > 160 * 1024) continue;
        out->T = T; out->W = W; out->G = G; out->shm = shm; out->sq = sq; out->hc = hc; out->s8 = s8; out->gr = gr; out->hg = hg; out->npl = npl; out->tg = tg;
        return true;
    }
    return false;
}

// ---- the cost model behind the choice of form: ONE table, every entry with the measurement it was fitted to -----------------------
// us per iteration of ONE round of the resident kernel (a round holds 8 x floor(32 / G) right-hand sides: one team per G CUs of an XCD)
//     t = a + g * G + s * T * (sites per lane / 4 where the form scales with them)
// and of the two-kernel streaming iteration (HBM-bound, linear in the batch).  Fitted on MI355X; a form that is not in the table is
// priced as the lane program.  The rule only has to order the two forms: a 20 % error in an entry moves a crossover by a few
// right-hand sides.
struct WgCost { const char *form; double a, g, s; const char *measured; };
static const WgCost wg_cost_table[] = {
    {"lane program (FORM 0)",            2.00, 0.12, 0.50, "profiles/r02/time_forms.log: configs B, C, D, E at 1-2 slices per wave; x sites per lane; + 3.9 for SSH at 2 slices, 0.85 T per slice from 4"},
    {"16x16 DPP (FORM 1, Holstein)",     2.66, 0.01, 0.62, "profiles/r03/time_forms.log: 3.5 / 4.0 / 5.2 us at 1 / 2 / 4 slices per wave"},
    {"16x16 DPP, bond phonons",          2.50, 0.00, 1.30, "profiles/r03/time_forms_E_ssh_dpp.log: 3.8 / 5.1 us at 1 / 2 slices"},
    {"honeycomb 12x12 DPP (FORM 2)",     2.10, 0.00, 1.70, "profiles/r03/time_forms_D_honeycomb_dpp.log: 3.8 / 5.5 us at 1 / 2 slices; 8.6 us at 3 (time_forms_D_three_slices_per_wave.log)"},
    {"8x8 DPP (FORM 4), team of one",    2.30, 0.00, 0.00, "profiles/r03/time_wg_B_8x8_dpp_form.log: 2.04-2.3 us whatever the batch"},
    {"GRID (FORM 5)",                    2.90, 0.01, 1.35, "profiles/r04/grid_form_even_L_square_lattices.log: L = 12: 4.3 / 5.6 / 10.0 us at 1 / 2 / 4 slices; L = 14, 10: 3.6-4.0 at 1"},
    {"TGRID (FORM 7, triangular)",       3.20, 0.00, 2.20, "profiles/r04/tgrid_triangular_resident.log: L = 16: 4.9 / 6.7 / 12.4 us at 1 / 2 / 4 slices per wave (4: 69 doubles spilled)"},
    {"HGRID (FORM 6)",                   2.30, 0.01, 1.10, "profiles/r04/hgrid_form_honeycomb_lattices.log: L = 6: 2.5 us (2 registers per lane), L = 10: 3.3-3.8 (4), L = 16: 4.5-5.8 (8); x sites per lane / 4"},
};
static const double wg_streaming_cost[3] = {10.0, 0.56, 0.02};   // us: launch floor + nrhs x (0.56 x Ndim / 40960 [x 1.1 for SSH] + 0.02) — profiles/r03/time_forms.log (C: 37 us at 64, 128 at 256)

static double resident_iteration_us(const elph_handle_s *h, const Shape &sh) {
    const bool ssh = h->kind == ELPH_MODEL_SSH;
    if (sh.s8) return wg_cost_table[4].a;
    if (sh.sq && ssh) return wg_cost_table[2].a + wg_cost_table[2].s * sh.T;
    if (sh.hc) return (sh.T == 3) ? 8.6 : wg_cost_table[3].a + wg_cost_table[3].s * sh.T;
    if (sh.sq) return wg_cost_table[1].a + wg_cost_table[1].g * sh.G + wg_cost_table[1].s * sh.T;
    if (sh.gr && sh.tg) return wg_cost_table[6].a + wg_cost_table[6].g * sh.G + wg_cost_table[6].s * sh.T;
    if (sh.gr) return wg_cost_table[5].a + wg_cost_table[5].g * sh.G + wg_cost_table[5].s * sh.T;
    if (sh.hg) return wg_cost_table[7].a + wg_cost_table[7].g * sh.G + wg_cost_table[7].s * sh.T * (sh.npl / 4.0);
    return wg_cost_table[0].a + wg_cost_table[0].g * sh.G + (sh.T >= 4 ? 0.85 : wg_cost_table[0].s) * sh.T * h->npl + ((ssh && sh.T == 2) ? 3.9 : 0.0);
}
static double streaming_iteration_us(const elph_handle_s *h, int nrhs) {
    return wg_streaming_cost[0] + nrhs * (wg_streaming_cost[1] * (double)h->ndim / 40960.0 * (h->kind == ELPH_MODEL_SSH ? 1.1 : 1.0) + wg_streaming_cost[2]);
}

template <int NPL, int T, bool SSH, bool UNI, int FORM, bool SHARD = false, bool X0Z = false>
static hipError_t launch_k(elph_handle_s *h, const Shape &sh, dim3 grid, const CgBufs &B, const ModelDev &m, const WgCtl &R,
                           const ShardCtl &Sh = ShardCtl()) {
    hipError_t e = hipFuncSetAttribute((const void *)k_cg_wg<NPL, T, SSH, UNI, FORM, SHARD, X0Z>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh.shm);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_cg_wg<NPL, T, SSH, UNI, FORM, SHARD, X0Z>), grid, dim3(sh.W * WAVE), sh.shm, h->stream, B, m, R, Sh);
    return hipGetLastError();
}

template <int NPL>
static hipError_t launch_shard_npl(elph_handle_s *h, const Shape &sh, dim3 grid, const CgBufs &B, const ModelDev &m, const WgCtl &R,
                                   const ShardCtl &Sh) {
    if (h->kind == ELPH_MODEL_SSH) return launch_k<NPL, 1, true, false, 0, true>(h, sh, grid, B, m, R, Sh);
    return m.uniform ? launch_k<NPL, 1, false, true, 0, true>(h, sh, grid, B, m, R, Sh)
                     : launch_k<NPL, 1, false, false, 0, true>(h, sh, grid, B, m, R, Sh);
}
// a shard in a register-exchange form: the slab is a periodic rectangle (its rows closed into a ring by the caller's bond table)
static hipError_t launch_shard_grid(elph_handle_s *h, const Shape &sh, dim3 grid, const CgBufs &B, const ModelDev &m, const WgCtl &R,
                                    const ShardCtl &Sh) {
    if (sh.gr) return launch_k<4, 1, false, true, 5, true>(h, sh, grid, B, m, R, Sh);
    switch (sh.npl) {
        case 2: return launch_k<2, 1, false, true, 6, true>(h, sh, grid, B, m, R, Sh);
        case 4: return launch_k<4, 1, false, true, 6, true>(h, sh, grid, B, m, R, Sh);
        default: return launch_k<8, 1, false, true, 6, true>(h, sh, grid, B, m, R, Sh);
    }
}

// the row form (k_cg_row) of the 4-slices-per-wave shape of the 16 x 16 DPP form: 32 slices per workgroup
static bool row_form(const elph_handle_s *h, const Shape &sh) {
    const char *e = getenv("ELPH_WG_ROW");
    if (!(e && e[0] == '1')) return false;
    return sh.sq && sh.T == 4 && sh.W == 8 && h->N == 256 && h->L % 32 == 0 && sh.G == (int)(h->L / 32) && sh.G >= 2 && sh.G <= 32;
}

template <int NPL>
static hipError_t launch_npl(elph_handle_s *h, const Shape &sh, dim3 grid, const CgBufs &B, const ModelDev &m, const WgCtl &R) {
    if constexpr (NPL == 4) {
        if (sh.sq) {
            if (h->kind == ELPH_MODEL_SSH) return (sh.T == 2) ? launch_k<4, 2, true, false, 1>(h, sh, grid, B, m, R) : launch_k<4, 1, true, false, 1>(h, sh, grid, B, m, R);
            if (!m.uniform) return (sh.T == 2) ? launch_k<4, 2, false, false, 1>(h, sh, grid, B, m, R) : launch_k<4, 1, false, false, 1>(h, sh, grid, B, m, R);
            if (sh.T == 4 && row_form(h, sh)) {      // the row form of the same shape (k_cg_row: a time slice per 16-lane row, 4 x 4 patches)
                const size_t shm = ((size_t)(34 + 34) * rowf::SV + 2 * rowf::SV + 2 * rowf::SV + 48) * sizeof(double);
                if (R.x0_zero) {
                    hipError_t e = hipFuncSetAttribute((const void *)k_cg_row<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
                    if (e != hipSuccess) return e;
                    hipLaunchKernelGGL((k_cg_row<true>), grid, dim3(8 * WAVE), shm, h->stream, B, m, R);
                } else {
                    hipError_t e = hipFuncSetAttribute((const void *)k_cg_row<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
                    if (e != hipSuccess) return e;
                    hipLaunchKernelGGL((k_cg_row<false>), grid, dim3(8 * WAVE), shm, h->stream, B, m, R);
                }
                return hipGetLastError();
            }
            if (sh.T == 4) return R.x0_zero ? launch_k<4, 4, false, true, 1, false, true>(h, sh, grid, B, m, R) : launch_k<4, 4, false, true, 1>(h, sh, grid, B, m, R);
            if (sh.T == 2) return launch_k<4, 2, false, true, 1>(h, sh, grid, B, m, R);
            return launch_k<4, 1, false, true, 1>(h, sh, grid, B, m, R);
        }
    }
    if (h->kind == ELPH_MODEL_SSH) {
        if constexpr (NPL <= 4) { if (sh.T == 2) return launch_k<NPL, 2, true, false, 0>(h, sh, grid, B, m, R); }
        return launch_k<NPL, 1, true, false, 0>(h, sh, grid, B, m, R);
    }
    if constexpr (NPL == 1) {
        if (sh.s8) {                // (the 8 x 8 DPP form)
            switch (sh.T) {
                case 8: return launch_k<1, 8, false, true, 4>(h, sh, grid, B, m, R);
                case 5: return launch_k<1, 5, false, true, 4>(h, sh, grid, B, m, R);
                case 4: return launch_k<1, 4, false, true, 4>(h, sh, grid, B, m, R);
                case 2: return launch_k<1, 2, false, true, 4>(h, sh, grid, B, m, R);
                default: return launch_k<1, 1, false, true, 4>(h, sh, grid, B, m, R);
            }
        }
    }
    if constexpr (NPL == 1) {       // (one workgroup per right-hand side: pick_shape)
        if (sh.T == 4) return m.uniform ? launch_k<1, 4, false, true, 0>(h, sh, grid, B, m, R) : launch_k<1, 4, false, false, 0>(h, sh, grid, B, m, R);
        if (sh.T == 5) return m.uniform ? launch_k<1, 5, false, true, 0>(h, sh, grid, B, m, R) : launch_k<1, 5, false, false, 0>(h, sh, grid, B, m, R);
        if (sh.T == 8) return m.uniform ? launch_k<1, 8, false, true, 0>(h, sh, grid, B, m, R) : launch_k<1, 8, false, false, 0>(h, sh, grid, B, m, R);
    }
    if constexpr (NPL <= 5) {
        if (sh.T == 2) return m.uniform ? launch_k<NPL, 2, false, true, 0>(h, sh, grid, B, m, R) : launch_k<NPL, 2, false, false, 0>(h, sh, grid, B, m, R);
    }
    return m.uniform ? launch_k<NPL, 1, false, true, 0>(h, sh, grid, B, m, R) : launch_k<NPL, 1, false, false, 0>(h, sh, grid, B, m, R);
}

}  // namespace wg

// Whether the workgroup-resident kernel can run this handle's un-preconditioned solves (and with which shape).
// How many workgroups of the resident kernels may a launch count on being resident AT ONCE on this handle's device?  Their teams spin on
// one another's records: a workgroup that is not resident is waited for until the time-out.  One workgroup per CU (the kernels run at 256
// registers x 8 waves, or need most of the LDS), a sixteenth of the CUs left to whatever else the device runs: 240 on a whole MI355X
// (256 CUs), proportionally fewer on a partitioned or CU-masked device (CPX: 32 CUs -> 30).  Queried once per device.
int elph_i_resident_wg_limit(const elph_handle_s *h) {
    static int cache[64] = {};
    const int d = (h && h->device >= 0 && h->device < 64) ? h->device : 0;
    if (cache[d] <= 0) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess || cus <= 0) cus = 256;
        cache[d] = std::max(1, cus - cus / 16);
    }
    return cache[d];
}

bool elph_wg_usable(const elph_handle_s *h, int *T, int *W, int *G, int nrhs) {
    const char *eo = getenv("ELPH_NO_WG");                 // read per call: the tests switch between the two forms
    const bool off = eo && eo[0] == '1';
    // (h->npl > 5: the lane-program form's limit — 320 sites; the honeycomb grid form carries up to 512 in one wave)
    const bool tri_grid = h->kind == ELPH_MODEL_HOLSTEIN && h->pg_kind == 3 && h->pg_L <= 16;      // (a six-colour lattice, but its resident form needs no lane program)
    if (off || !h->fast || (h->lp_mc != 4 && !tri_grid) || (h->npl > 5 && !(h->hc_LX > 0 && !h->hc12 && h->kind == ELPH_MODEL_HOLSTEIN)) || h->dot_hi != 0 || h->solo_chain >= 0) return false;
    const char *et = getenv("ELPH_WG_T");
    wg::Shape sh;
    if (!wg::pick_shape(h, elph_model_dev(h), et ? atoi(et) : 0, nrhs, &sh)) return false;
    if (T) *T = sh.T;
    if (W) *W = sh.W;
    if (G) *G = sh.G;
    return true;
}

// A team timed out on an earlier solve (the GPU was shared with something that held its CUs): the handle runs the streaming iteration
// for the next ELPH_WG_COOLDOWN solves (default 16), then tries the resident kernels again with a cleared abort word.  One step per
// solve that WOULD have taken a resident kernel — called by both of them (elph_wg_cg, elph_pcg_wg), so a handle that only ever runs
// preconditioned solves recovers too.
int elph_wg_cooldown_step(elph_handle_s *h) {
    if (!h->wg_broken) return ELPH_OK;
    if (--h->wg_cooldown > 0) return ELPH_OK;
    h->wg_broken = false;
    if (h->d_res && h->wg_abort_off) HIPCHK(hipMemsetAsync(static_cast<char *>(h->d_res) + h->wg_abort_off, 0, sizeof(int), h->stream));
    return ELPH_OK;
}

// Runs the whole un-preconditioned CG for rhs [0, nrhs) after elph_launch_cg_init (fixed_iters > 0: exactly that many
// iterations without stop test — measurement).  *ran = false: not applicable, nothing was launched.
// ELPH_E_HIP with "workgroup-resident" in the message: a team timed out; the caller re-initialises and runs the two-kernel path.
int elph_wg_cg(elph_handle_s *h, const CgBufs &B, int nrhs, long long fixed_iters, bool *ran) {
    *ran = false;
    if (B.params.use_prec) return ELPH_OK;
    if (h->wg_broken) return ELPH_OK;                   // (cooling down after a time-out: elph_wg_cooldown_step, run_cg)
    if (!elph_wg_usable(h, nullptr, nullptr, nullptr, nrhs)) return ELPH_OK;
    const char *et = getenv("ELPH_WG_T");
    wg::Shape sh;
    ModelDev m = elph_model_dev(h);
    if (!wg::pick_shape(h, m, et ? atoi(et) : 0, nrhs, &sh)) return ELPH_OK;
    // Which form is faster for THIS batch: a deterministic rule (never a timing at run time — which form runs decides the last bits of a
    // solution) from ONE table of measured constants, wg_cost_table above.  fixed_iters > 0 (measurement of this kernel) and
    // ELPH_WG_ALWAYS=1 skip it.
    {
        const char *ea = getenv("ELPH_WG_ALWAYS");
        if (fixed_iters <= 0 && !(ea && ea[0] == '1')) {
            const int per_round = 8 * std::max(1, 32 / sh.G);
            const double rounds = (double)((nrhs + per_round - 1) / per_round);
            if (rounds * wg::resident_iteration_us(h, sh) > wg::streaming_iteration_us(h, nrhs)) return ELPH_OK;
        }
    }
    const size_t HS = (size_t)sh.npl * WAVE;
    const size_t n_slots = 2 * (size_t)nrhs * wg::SLOTS_PER_RHS, n_bnd = (sh.G > 1) ? 2 * (size_t)nrhs * sh.G * 2 * HS * 2 : 0;      // (x 2: by the parity of the iteration)
    const size_t need = (n_slots + n_bnd) * sizeof(wg::u64) + 64;
    // tags: a range of (iterations + 2) values per launch; the control block is zeroed only when it is (re)allocated or the
    // 32-bit range wraps.  The abort word sits at the END of the allocation (its place must not move with the batch size).
    const unsigned long long span = (unsigned long long)std::min<long long>(fixed_iters > 0 ? fixed_iters : B.params.maxiter, 1LL << 30) + 2;
    bool zero = false;
    if (need > h->res_cap) {
        HIPCHK(hipStreamSynchronize(h->stream));
        if (h->d_res) HIPCHK(hipFree(h->d_res));
        h->d_res = nullptr;
        HIPCHK(hipMalloc(&h->d_res, need));
        h->res_cap = need;
        zero = true;
    }
    if ((unsigned long long)h->wg_epoch + span >= 0xFFFFFFFFull) zero = true;
    if (zero) { HIPCHK(hipMemsetAsync(h->d_res, 0, h->res_cap, h->stream)); h->wg_epoch = 0; }
    wg::WgCtl R;
    char *base = static_cast<char *>(h->d_res);
    R.slots = reinterpret_cast<wg::u64 *>(base);
    R.bnd = R.slots + n_slots;
    R.abort = reinterpret_cast<int *>(base + h->res_cap - 64);
    R.epoch0 = h->wg_epoch;
    h->wg_epoch += (unsigned)span;
    R.G = sh.G; R.W = sh.W;
    const char *eto = getenv("ELPH_WG_TIMEOUT_MS");
    // (2 s: a team at the dispatch frontier of an oversubscribed grid waits for whole solves of the resident ones — tens of ms for a
    //  batch of hundreds of right-hand sides; a measurement launch of thousands of fixed iterations gets its own duration on top)
    R.timeout_ticks = ((long long)(eto ? atoll(eto) : 2000) + (fixed_iters > 0 ? fixed_iters / 10 : 0)) * 100000LL;     // wall_clock64 runs at 100 MHz
    R.fixed_iters = fixed_iters;
    { const char *ez = getenv("ELPH_WG_X0Z"); R.x0_zero = (h->wg_x0_zero && !(ez && ez[0] == '0')) ? 1 : 0; }     // (x0 = 0 known: the 4-slice DPP shape has an instantiation that does not read it)
    h->wg_x0_zero = false;
    if (fixed_iters <= 0)       // the caller's initial guess survives in d_zp (unused by an un-preconditioned solve) for the fallback
        HIPCHK(hipMemcpyAsync(h->d_zp, h->d_x, (size_t)nrhs * (size_t)h->ndim * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
#ifdef ELPH_WG_PERSISTENT
    // persistent teams: per XCD (32 CUs, one workgroup each) floor(32 / G) teams at most
    R.teams_per_xcd = std::max(1, std::min((nrhs + 7) / 8, 32 / sh.G));
#else
    R.teams_per_xcd = (nrhs + 7) / 8;                      // one solve per workgroup: every right-hand side has its team in the grid
#endif
    const dim3 grid((unsigned)(8 * R.teams_per_xcd * sh.G));
    hipError_t e = hipSuccess;
    if (sh.hg) {
        switch (sh.npl) {
            case 2: e = (sh.T == 2) ? wg::launch_k<2, 2, false, true, 6>(h, sh, grid, B, m, R) : wg::launch_k<2, 1, false, true, 6>(h, sh, grid, B, m, R); break;
            case 4: e = (sh.T == 2) ? wg::launch_k<4, 2, false, true, 6>(h, sh, grid, B, m, R) : wg::launch_k<4, 1, false, true, 6>(h, sh, grid, B, m, R); break;
            default: e = (sh.T == 2) ? wg::launch_k<8, 2, false, true, 6>(h, sh, grid, B, m, R) : wg::launch_k<8, 1, false, true, 6>(h, sh, grid, B, m, R); break;
        }
    } else if (sh.gr && sh.tg) {
        e = (sh.T == 4) ? wg::launch_k<4, 4, false, true, 7>(h, sh, grid, B, m, R)
          : (sh.T == 2) ? wg::launch_k<4, 2, false, true, 7>(h, sh, grid, B, m, R) : wg::launch_k<4, 1, false, true, 7>(h, sh, grid, B, m, R);
    } else if (sh.gr) {
        e = (sh.T == 4) ? wg::launch_k<4, 4, false, true, 5>(h, sh, grid, B, m, R)
          : (sh.T == 2) ? wg::launch_k<4, 2, false, true, 5>(h, sh, grid, B, m, R) : wg::launch_k<4, 1, false, true, 5>(h, sh, grid, B, m, R);
    } else if (sh.hc) {
        e = (sh.T == 3) ? wg::launch_k<wg::HC_NPL, 3, false, true, 2>(h, sh, grid, B, m, R)
          : (sh.T == 2) ? wg::launch_k<wg::HC_NPL, 2, false, true, 2>(h, sh, grid, B, m, R) : wg::launch_k<wg::HC_NPL, 1, false, true, 2>(h, sh, grid, B, m, R);
    } else switch (h->npl) {
        case 1: e = wg::launch_npl<1>(h, sh, grid, B, m, R); break;
        case 2: e = wg::launch_npl<2>(h, sh, grid, B, m, R); break;
        case 3: e = wg::launch_npl<3>(h, sh, grid, B, m, R); break;
        case 4: e = wg::launch_npl<4>(h, sh, grid, B, m, R); break;
        default: e = wg::launch_npl<5>(h, sh, grid, B, m, R); break;
    }
    if (e != hipSuccess) { elph_set_error("launch k_cg_wg failed: %s", hipGetErrorString(e)); return ELPH_E_HIP; }
    h->wg_T = sh.T; h->wg_W = sh.W; h->wg_G = sh.G;
    h->wg_abort_off = h->res_cap - 64;
    *ran = true;
    return ELPH_OK;
}

// after the stream has drained: did a team give up?  (the abort word follows the records in the control block)
int elph_wg_aborted(elph_handle_s *h, bool *aborted) {
    *aborted = false;
    if (!h->d_res || h->wg_abort_off == 0) return ELPH_OK;
    int ab = 0;
    HIPCHK(hipMemcpy(&ab, static_cast<char *>(h->d_res) + h->wg_abort_off, sizeof(int), hipMemcpyDeviceToHost));
    if (ab) {
        h->wg_broken = true;
        const char *ec = getenv("ELPH_WG_COOLDOWN");
        h->wg_cooldown = ec ? std::max(1, atoi(ec)) : 16;
        ++h->wg_fallbacks;
        *aborted = true;
        elph_set_error("workgroup-resident CG timed out waiting for its team (T=%d W=%d G=%d); falling back to the two-kernel iteration",
                       h->wg_T, h->wg_W, h->wg_G);
    }
    return ELPH_OK;
}

#ifdef ELPH_WG_ARRIVE
extern "C" int elph_debug_wg_arrive(unsigned long long *out, int n_blocks, int clear) {
    if (clear) { static unsigned long long z[4096 * 4]; return hipMemcpyToSymbol(HIP_SYMBOL(wg::g_wg_arrive), z, sizeof(z)) == hipSuccess ? 0 : -1; }
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(wg::g_wg_arrive), (size_t)n_blocks * 4 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif
#ifdef ELPH_WG_ARRIVE
extern "C" int elph_debug_wg_timeline(unsigned long long *out, int n_blocks) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(wg::g_wg_timeline), (size_t)n_blocks * 4 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif
#ifdef ELPH_WG_STAMPS
extern "C" int elph_debug_wg_stamps(unsigned long long *out16) {
    return hipMemcpyFromSymbol(out16, HIP_SYMBOL(wg::g_wg_stamps), 16 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif

// One rank's launch of a solve over several GPUs (shard.hip holds the mailbox and calls this).  x0 = 0, b in B.r and B.p.
// one rank's launch arguments of a sharded solve: form and team shape of its slab, control block (zeroed), tags
static int shard_rank_setup(elph_handle_s *h, const CgBufs &B, long long fixed_iters, const ElphShardCtl &Sh, ModelDev &m, wg::Shape &sh,
                            wg::WgCtl &R) {
    m = elph_model_dev(h);
    // a slab whose rows the caller closed into a ring (a periodic rectangle in the reference's colouring: sharded.py, ring=True) runs a
    // register-exchange form — the own rows of z = M^T M p do not see the ring bond (it lies beyond the ghost rows' dependency closure)
    // (measured, profiles/r04/shard_ring_grid_forms_ab.log: the GRID form pays from three sites per lane of the lane program — a
    //  slab of up to 128 sites is two LDS values per lane and colour, cheaper than four registers on half the lanes: config C over 4 / 8
    //  ranks 5.3 / 6.0 us in the lane program against 6.5 / 7.2; the HGRID form with 8 registers per lane spills in the shard's
    //  kernel: config D on one rank 16.6 against 9.6 us)
    const bool gr = wg::gr_form(h, m, true) && h->npl >= 3;
    int hgn = gr ? 0 : wg::hg_form(h, m, true);
    if (hgn == 8) hgn = 0;
    if (!gr && !hgn && (!h->fast_capable || h->lp_mc != 4 || h->npl > 5)) {
        elph_set_error("sharded solve: the slab needs a 4-colour lane program and <= 320 sites (N = %lld)", (long long)h->N);
        return ELPH_E_UNSUPPORTED;
    }
    {   // one slice per wave on every rank (slab sizes differ between ranks; the team shape must not): elph_shard_shape
        int W = 0, G = 0;
        const int rc = elph_shard_shape(h->L, Sh.P, &W, &G, nullptr, nullptr);
        if (rc) return rc;
        sh.npl = gr ? 4 : (hgn ? hgn : h->npl);
        const size_t SL = (size_t)sh.npl * WAVE + 2 * WAVE, HS = (size_t)sh.npl * WAVE;
        sh.T = 1; sh.W = W; sh.G = G; sh.sq = false; sh.hc = false; sh.s8 = false; sh.gr = gr; sh.hg = hgn > 0;
        sh.shm = ((size_t)W * ((gr || hgn) ? 0 : 2) * SL + 2 * (size_t)W * HS + 48 + 4 * HS) * sizeof(double);     // + partials, totals, rhalo[2][HS], zhalo[2][HS]
    }
    const size_t HS = (size_t)sh.npl * WAVE;
    const size_t n_slots = 2 * wg::SLOTS_PER_RHS, n_bnd = (sh.G > 1) ? 2 * (size_t)sh.G * 2 * HS * 2 : 0;
    const size_t need = (n_slots + n_bnd) * sizeof(wg::u64) + 64;
    // tags: a range of (iterations + 2) values per launch; the control block is zeroed only when it is (re)allocated or the
    // 32-bit range wraps.  The abort word sits at the END of the allocation (its place must not move with the batch size).
    const unsigned long long span = (unsigned long long)std::min<long long>(fixed_iters > 0 ? fixed_iters : B.params.maxiter, 1LL << 30) + 2;
    bool zero = false;
    if (need > h->res_cap) {
        HIPCHK(hipStreamSynchronize(h->stream));
        if (h->d_res) HIPCHK(hipFree(h->d_res));
        h->d_res = nullptr;
        HIPCHK(hipMalloc(&h->d_res, need));
        h->res_cap = need;
        zero = true;
    }
    if ((unsigned long long)h->wg_epoch + span >= 0xFFFFFFFFull) zero = true;
    if (zero) { HIPCHK(hipMemsetAsync(h->d_res, 0, h->res_cap, h->stream)); h->wg_epoch = 0; }
    char *base = static_cast<char *>(h->d_res);
    R.slots = reinterpret_cast<wg::u64 *>(base);
    R.bnd = R.slots + n_slots;
    R.abort = reinterpret_cast<int *>(base + h->res_cap - 64);
    R.epoch0 = 0;                                          // (the records of a sharded solve live in the ranks' mailboxes: one numbering for all)
    (void)span;
    R.G = sh.G; R.W = sh.W;
    // (a sharded solve has NO streaming fallback behind it — a time-out is a failed solve — and its ranks start with whatever skew the
    //  host's barrier, first-launch code loading and time-slicing leave: the long bound, elph_shard_timeout_ms)
    R.timeout_ticks = elph_shard_timeout_ms() * 100000LL;
    R.fixed_iters = fixed_iters;
    R.x0_zero = 0;
    HIPCHK(hipMemsetAsync(base, 0, h->res_cap, h->stream));   // boundary granules of this rank's workgroups: tags restart at 2
    h->wg_epoch = 0;
    R.teams_per_xcd = 1;
    return ELPH_OK;
}

int elph_wg_cg_shard(elph_handle_s *h, const CgBufs &B, long long fixed_iters, const ElphShardCtl &Sh, int *G_out) {
    ModelDev m;
    wg::Shape sh;
    wg::WgCtl R;
    const int rcs = shard_rank_setup(h, B, fixed_iters, Sh, m, sh, R);
    if (rcs) return rcs;
    const dim3 grid((unsigned)sh.G);                       // one right-hand side: its G workgroups, round-robin over the XCDs
    hipError_t e = hipSuccess;
    if (sh.gr || sh.hg) e = wg::launch_shard_grid(h, sh, grid, B, m, R, Sh);
    else switch (h->npl) {
        case 1: e = wg::launch_shard_npl<1>(h, sh, grid, B, m, R, Sh); break;
        case 2: e = wg::launch_shard_npl<2>(h, sh, grid, B, m, R, Sh); break;
        case 3: e = wg::launch_shard_npl<3>(h, sh, grid, B, m, R, Sh); break;
        case 4: e = wg::launch_shard_npl<4>(h, sh, grid, B, m, R, Sh); break;
        default: e = wg::launch_shard_npl<5>(h, sh, grid, B, m, R, Sh); break;
    }
    if (e != hipSuccess) { elph_set_error("launch k_cg_wg (shard) failed: %s", hipGetErrorString(e)); return ELPH_E_HIP; }
    h->wg_T = sh.T; h->wg_W = sh.W; h->wg_G = sh.G;
    h->wg_abort_off = h->res_cap - 64;
    if (G_out) *G_out = sh.G;
    return ELPH_OK;
}

// ALL RANKS OF A SHARDED SOLVE WHOSE SLABS LIVE ON ONE DEVICE, IN ONE LAUNCH (slabs.hip): the same kernel, rank q's workgroups = blocks
// q G .. q G + G - 1, launch arguments from d_args (device memory, P x elph_wg_rank_args_bytes()).  Every slab must take the lane-program
// form with the same sites per lane and the same team shape; all P G workgroups must be resident at once (they wait for each other).
size_t elph_wg_rank_args_bytes() { return sizeof(wg::WgRankArgs); }

int elph_wg_cg_ranks(elph_handle_s *const *hs, int P, const CgBufs *Bs, long long fixed_iters, const ElphShardCtl *ctls, void *h_args,
                     void *d_args, hipStream_t stream, long long timeout_ms, int *G_out) {
    if (P < 1 || P > 2 * ELPH_SHARD_MAXRANKS) { elph_set_error("bad rank count %d", P); return ELPH_E_ARG; }      // (up to two sets of slabs)
    wg::WgRankArgs *A = static_cast<wg::WgRankArgs *>(h_args);          // (pinned, owned by the caller: the copy below is asynchronous)
    wg::Shape sh0;
    bool uni = true;
    for (int q = 0; q < P; ++q) {
        if (hs[q]->stream != stream) { elph_set_error("slab %d runs on another stream", q); return ELPH_E_STATE; }
        wg::Shape sh;
        const int rc = shard_rank_setup(hs[q], Bs[q], fixed_iters, ctls[q], A[(size_t)q].m, sh, A[(size_t)q].R);
        if (rc) return rc;
        if (sh.hg || hs[q]->kind != ELPH_MODEL_HOLSTEIN) { elph_set_error("slabs on one device: lane-program or GRID form, site phonons"); return ELPH_E_UNSUPPORTED; }
        if (q == 0) sh0 = sh;
        else if (sh.npl != sh0.npl || sh.W != sh0.W || sh.G != sh0.G || sh.shm != sh0.shm || sh.gr != sh0.gr) { elph_set_error("slabs on one device: slab %d has another shape", q); return ELPH_E_UNSUPPORTED; }
        uni = uni && A[(size_t)q].m.uniform;
        if (timeout_ms > 0) A[(size_t)q].R.timeout_ticks = timeout_ms * 100000LL;      // (slabs of one device: a streaming fallback exists, the short bound)
        A[(size_t)q].B = Bs[q];
        A[(size_t)q].Sh = ctls[q];
    }
    if ((long long)P * sh0.G > elph_i_resident_wg_limit(hs[0])) { elph_set_error("slabs on one device: %d x %d workgroups cannot all be resident", P, sh0.G); return ELPH_E_UNSUPPORTED; }
    HIPCHK(hipMemcpyAsync(d_args, A, (size_t)P * sizeof(wg::WgRankArgs), hipMemcpyHostToDevice, stream));
    wg::WgCtl R0 = A[0].R;
    R0.ranks = d_args;
    // (tests, ELPH_SLABS_TEST_TIMEOUT=2: the last rank's workgroups are NOT launched — the others wait for them until their bound and give up)
    const char *edrop = getenv("ELPH_SLABS_TEST_TIMEOUT");
    const int drop = (edrop && edrop[0] == '2') ? 1 : 0;
    const dim3 grid((unsigned)((P - drop) * sh0.G)), block((unsigned)(sh0.W * WAVE));
    hipError_t e = hipSuccess;
#define RANKS_LAUNCH(NPLV, UNIV) do {                                                                                                  \
        auto kfn = wg::k_cg_wg<NPLV, 1, false, UNIV, 0, true, false, true>;                                                            \
        e = hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh0.shm);                          \
        if (e == hipSuccess) { hipLaunchKernelGGL(kfn, grid, block, sh0.shm, stream, Bs[0], A[0].m, R0, ctls[0]); e = hipGetLastError(); } \
    } while (0)
#define RANKS_CASE(NPLV) case NPLV: if (uni) RANKS_LAUNCH(NPLV, true); else RANKS_LAUNCH(NPLV, false); break;
    if (sh0.gr) {      // slabs closed into rings: periodic rectangles in the reference's colouring, the GRID form (2 x 2 patches per lane)
        auto kfn = wg::k_cg_wg<4, 1, false, true, 5, true, false, true>;
        e = hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh0.shm);
        if (e == hipSuccess) { hipLaunchKernelGGL(kfn, grid, block, sh0.shm, stream, Bs[0], A[0].m, R0, ctls[0]); e = hipGetLastError(); }
    } else
    switch (sh0.npl) {
        RANKS_CASE(1) RANKS_CASE(2) RANKS_CASE(3) RANKS_CASE(4)
        default: if (uni) RANKS_LAUNCH(5, true); else RANKS_LAUNCH(5, false); break;
    }
#undef RANKS_CASE
#undef RANKS_LAUNCH
    if (e != hipSuccess) { elph_set_error("launch k_cg_wg (ranks) failed: %s", hipGetErrorString(e)); return ELPH_E_HIP; }
    for (int q = 0; q < P; ++q) { hs[q]->wg_T = 1; hs[q]->wg_W = sh0.W; hs[q]->wg_G = sh0.G; hs[q]->wg_abort_off = hs[q]->res_cap - 64; }
    if (G_out) *G_out = sh0.G;
    return ELPH_OK;
}
