#!/usr/bin/env python3
"""bench.py — CG mat-vecs/s on the L=16, Ntau=160 Holstein square lattice (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--nrhs R] [--precond]

A *step* is one conjugate-gradient iteration (1 MtM apply = 2 mat-vecs, + the vector updates and the reductions; + 1 KPM apply with
--precond) advanced for a batch of `nrhs` right-hand sides resident in HBM.  Default batch: 144 independent Markov chains per GPU
(144 phonon configurations = 144 different fermion matrices; the reference runs them as separate processes, ElPhDynamics.jl:90-95)
x the 2 pseudofermion solves of one HMC force evaluation each (HMC.jl:851-886) = 288 right-hand sides.  Un-preconditioned solves
run as the workgroup-resident kernel (csrc/cg_wg.hip: the whole solve in one launch), so the K timed steps are ONE launch of it.
W untimed warm-up steps, then exactly K steps bracketed by barrier + device synchronise on both sides; the time is the MAX over
ranks and value = (2 * nrhs * K * n_gpus) / time.  One JSON line on rank 0.

N > 1 (launched by torch.distributed.run, one rank per GPU): every GPU carries its own chains — no data-path collective, "weak"
scaling (SURVEY.md §8e replica mode) — that is `value`.  The same run then records the north_star's OTHER curve as the sub-record
`spatial`: ONE solve of configs C, D and E sharded over the N ranks (row slabs + ghost rows, device-initiated mailbox stores, the
in-library path of csrc/shard.hip) with its device time per iteration, mat-vecs/s, fraction of the f64 roof of the N GPUs, the rank
count RCCL sees and the peer-access matrix.  `--mode spatial` makes that solve (of --config) the headline line instead: strong
scaling, latency-bound at these sizes; its line carries `roofline` and, on one rank, `cpu_baseline` too.

The JSON line (the driver's record keeps SCALARS of `roofline` and `cpu_baseline` only, so everything that matters is a flat key):
  roofline      the dominant kernel of the timed region.  Resident kernel: bound = "f64_vector+sync" — achieved = flops of the launch
                (2 mat-vecs x (2 Ndim + 6 Ltau Nbonds) + 10 Ndim per right-hand side and iteration) / launch duration
                against the 78.6 TFLOP/s f64 vector peak; its HBM fraction (small by design: the Krylov vectors stay on chip) is
                hbm_frac.  streaming_*: the two-kernel iteration at the same batch (k_cg_ap on its compulsory bytes against 8 TB/s).
                precond_*: the KPM-preconditioned iteration (BASELINE config "with tau-FFT precond"): every kernel timed alone with
                HIP events, priced on the compulsory bytes of ALL its kernels.
  cpu_baseline  the CPU oracle (oracle/elph_oracle.c, -O3 -march=native -ffast-math, 1 thread = the reference's configuration,
                ElPhDynamics.jl:74-75) on a bounded sample of the same workload.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md
HBM_STREAM_CEILING_GBS = 5380.0   # MEASURED on this pool (round 6): read r, read z, write r of 288 x 40 960 doubles each, buffers rotating through more than the
                                  # 256 MB Infinity Cache holds: 5.29-5.39 TB/s in three access patterns (tools/probes/access_pattern_probe.cpp; profiles/r06/)
HMC_CHAINS = 64                # lockstep chains of the secondary whole-HMC-update measurement
F64_MFMA_PEAK_TFLOPS = 78.6    # v_mfma_f64_16x16x4_f64, dense (same guide: f64 matrix = f64 vector peak)
ALG_BYTES_PER_ELT = {          # the UNFUSED pass count of the algorithm, f64 (reported as algorithmic_GBs only)
    "cg_iter": 120.0,          # 2 mat-vecs (48) + x, r, p updates (72)
    "k_cg_ap": 96.0,           # x += alpha p of the previous iteration (24) + p = r + beta p (24) + M p (24) + Mt (M p) (24)
    "k_cg_xr": 24.0,           # r -= alpha z (24)   [x += alpha p rides in the next k_cg_ap, which reads that p anyway]
    "kpm_apply": 16.0,
}
# COMPULSORY bytes of each kernel AS BUILT (what roofline.frac is priced on): every vector the kernel must read or write
# once, 8 B per element per right-hand side; tables that are shared by right-hand sides are counted once per launch.
#   k_cg_ap     reads src (r | P^-1 r), p_old, x; writes p, z, x                     -> 48 B/elt/rhs  (+ E once per chain; SSH:
#               + the per-slice hopping tables once per chain).  M p never leaves LDS, x += alpha p shares the read of p.
#   k_cg_xr     reads r, z; writes r                                                  -> 24 B/elt/rhs
#   KPM apply as three kernels: forward transform with the residual update folded in (reads r, z; writes r, nu) 32;
#               Chebyshev recursion in place on nu (read + write) 16; inverse transform (reads nu, writes P^-1 r) 16
BUILT_BYTES_PER_ELT = {"k_cg_ap": 48.0, "k_cg_xr": 24.0, "kpm_fwd_xr": 32.0, "kpm_fwd": 16.0, "kpm_cheb": 16.0, "kpm_inv": 16.0}


DESCR = {"A": "Holstein single site (holstein_hmc_single_site.toml)", "B": "Holstein square L=8 Ntau=40",
         "C": "Holstein square L=16 Ntau=160", "D": "Holstein honeycomb L=12 Ntau=120", "E": "optical SSH square L=16 Ntau=160"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4000)
    ap.add_argument("--warmup", type=int, default=400)
    ap.add_argument("--nrhs", type=int, default=288, help="right-hand sides advanced per step (batch)")
    ap.add_argument("--chains", type=int, default=144,
                    help="independent phonon configurations (Markov chains) per GPU sharing the batch: right-hand side r "
                         "uses the fermion matrix of chain r %% chains (nrhs = 2*chains = both pseudofermion solves of one "
                         "HMC force evaluation per chain); 1 = all right-hand sides on one matrix")
    ap.add_argument("--mode", default="chains", choices=["chains", "spatial"],
                    help="chains (default): independent chains per GPU, no data-path collective (weak scaling; with N > 1 the line "
                         "also carries the `spatial` sub-record).  spatial: ONE solve of --config over the N ranks, slabs of rows of "
                         "cells + ghost rows (the north_star's decomposition), device-initiated mailbox stores, as the headline "
                         "(strong scaling; latency-bound at these sizes)")
    ap.add_argument("--spatial-steps", type=int, default=2000, help="iterations of each sharded solve of the `spatial` sub-record")
    ap.add_argument("--ranks-per-proc", type=int, default=1,
                    help="--mode spatial only: rank threads per process (rehearsal of more ranks than the one-GPU box admits "
                         "processes: 8 ranks = 4 processes x 2; --gpus = total ranks)")
    ap.add_argument("--config", default="C", help="BASELINE config tag (C = Holstein square L=16 Ltau=160)")
    ap.add_argument("--precond", action="store_true", help="KPM (tau-FFT) preconditioned CG iteration")
    ap.add_argument("--streaming", action="store_true", help="time the two-kernel (HBM-streaming) iteration instead of the workgroup-resident kernel")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-sweep", action="store_true", help="skip the secondary nrhs sweep")
    ap.add_argument("--no-spatial", action="store_true", help="skip the `spatial` sub-record (sharded solves of C, D, E over the ranks)")
    ap.add_argument("--cpu-seconds", type=float, default=4.0)
    ap.add_argument("--hmc-seconds", type=float, default=6.0, help="wall time of the back-to-back HMC updates of the 64-chain leg")
    return ap.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) started WITHOUT a launcher: start the N rank processes ourselves — fresh children of
    `python -m torch.distributed.run` on 127.0.0.1, before this process has touched a GPU or imported torch (a child, never an exec) —
    and leave with their exit code.  A request for N GPUs therefore either prints a line with n_gpus = N or fails; it never prints a
    one-GPU line.  (The reference's way of filling a node is N independent run-IDs, ElPhDynamics.jl:90-95.)"""
    import socket
    import subprocess
    if os.environ.get("ELPH_BENCH_NO_SELF_LAUNCH") == "1":
        raise SystemExit(f"bench.py: --gpus {args.gpus} needs {args.gpus} ranks (WORLD_SIZE is not set and self-launch is disabled): "
                         f"python -m torch.distributed.run --nnodes=1 --nproc-per-node {args.gpus} --master-addr 127.0.0.1 bench.py --gpus {args.gpus} ...")
    rpp = max(1, args.ranks_per_proc if args.mode == "spatial" else 1)
    if args.gpus % rpp:
        raise SystemExit(f"--gpus {args.gpus} is not a multiple of --ranks-per-proc {rpp}")
    nproc = args.gpus // rpp
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"bench.py: --gpus {args.gpus} without a launcher: starting {nproc} rank process(es): {' '.join(cmd)}", file=sys.stderr, flush=True)
    env = dict(os.environ, ELPH_BENCH_SELF_LAUNCHED="1")
    raise SystemExit(subprocess.call(cmd, env=env))


def dry_line(args, comm):
    """ELPH_BENCH_DRY=1: the launch / rendezvous / reduction path of an N-rank run WITHOUT device work (the CPU test of self_launch at
    world 2 over gloo) — value is null and the line says so; never a measurement."""
    from elphdynamics_amd import dist as edist
    elapsed, work = edist.timed_steps(comm, lambda k: (time.sleep(0.01 * (1 + comm.rank)), 2.0 * args.nrhs * k)[1], max(1, args.steps))
    if comm.rank == 0:
        print(json.dumps({"metric": "cg_matvecs_per_sec", "value": None, "unit": "matvec/s", "n_gpus": comm.world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                          "dtype": "f64", "data": "DRY RUN: no device work (ELPH_BENCH_DRY=1), launch path only", "dry_run": True,
                          "work_all_ranks": work, "elapsed_max": elapsed, "self_launched": os.environ.get("ELPH_BENCH_SELF_LAUNCHED") == "1",
                          "dist_backend": comm.backend}))
    comm.close()


def main():
    args = parse()
    if args.gpus > 1 and int(os.environ.get("WORLD_SIZE", "1")) <= 1:
        self_launch(args)              # does not return
    rpp = max(1, args.ranks_per_proc) if args.mode == "spatial" else 1
    if args.gpus != int(os.environ.get("WORLD_SIZE", "1")) * rpp:      # before the rendezvous: a wrong rank count must not hang in it
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={os.environ.get('WORLD_SIZE', '1')} x {rpp} rank thread(s) per process")
    from elphdynamics_amd import dist as edist
    comm = edist.Comm()            # imports torch (and initialises RCCL) only when WORLD_SIZE > 1
    rank, local_rank, world = comm.rank, comm.local_rank, comm.world
    if args.mode == "spatial" and args.ranks_per_proc > 1:
        if args.gpus != world * args.ranks_per_proc:
            raise SystemExit(f"--gpus {args.gpus} but {world} process(es) x {args.ranks_per_proc} rank threads")
        edist.HybridComm.spawn(comm, args.ranks_per_proc, lambda hc: main_sharded(args, hc))
        comm.close()
        return
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if os.environ.get("ELPH_BENCH_DRY") == "1":
        return dry_line(args, comm)

    if world > 1:        # the CPU baseline and the secondary measurements belong to the N = 1 run (rank 0 only, contract ④)
        args.no_cpu = True
        args.no_sweep = True
    import numpy as np
    if args.mode == "spatial":
        return main_sharded(args, comm)
    from elphdynamics_amd import _lib, configs, models, preconditioners as pc, synth
    from elphdynamics_amd._lib import check

    lib = _lib.load()
    if lib.elph_device_count() < 1:
        raise SystemExit("bench.py needs an MI355X: libelphgpu has no CPU path")

    # one independent chain (phonon configuration) per rank
    m = configs.make_model(args.config, tol=1e-5, device=comm.device_index(),
                           seed=comm.chain_seed(synth.SEED_FIELDS))
    nrhs = args.nrhs
    nchains = max(1, min(args.chains, nrhs))
    R, B = configs.rhs(m, nrhs, seed=comm.chain_seed(synth.SEED_RHS))
    if nchains > 1:      # every chain its own phonon configuration (its own fermion matrix)
        if m.kind == 0:
            Xc = np.stack([synth.phonon_field(m.Nph, m.Ltau, m.beta, m.dtau, seed=comm.chain_seed(synth.SEED_FIELDS) + 17 * c)
                           for c in range(nchains)])
        else:            # bond phonons: the deck's field rescaled and roughened per chain (|alpha x| stays below t)
            Xc = np.stack([m.x * (0.6 + 0.8 * c / nchains) * (1.0 + 0.2 * synth.randn(comm.chain_seed(synth.SEED_FIELDS) + 17 * c, m.Ndof))
                           for c in range(nchains)])
        models.update_model_chains_(m, Xc)
    # un-preconditioned solves run as the workgroup-resident kernel (cg_wg.hip: the whole solve in one launch) wherever it applies —
    # that is what elph_ldiv / elph_cg_solve launch — so that is what the headline times: K iterations of every right-hand side of
    # the batch = ONE launch (what = 9).  --streaming times the two-kernel iteration instead (the form used for preconditioned
    # solves and for lattices the resident kernel does not take), one pair of launches per step.
    wg_us, wg_T, wg_W, wg_G = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    check(lib.elph_bench_wg_info(m._h, nrhs, C.byref(wg_us), C.byref(wg_T), C.byref(wg_W), C.byref(wg_G)))
    resident = bool(wg_us.value) and not args.precond and not args.streaming
    what = 3 if args.precond else (9 if resident else 1)
    # a KPM-preconditioned batch from 192 right-hand sides (64 on lattices of five and more sites per lane: config D) runs as two half-batches on two
    # streams (elph_api.hip: SplitRun, split_wanted): time that form
    split_from = 64 if (m.Nsites + 63) // 64 >= 5 else 192
    two_streams = bool(args.precond and nrhs >= split_from and os.environ.get("ELPH_SPLIT_STREAMS") != "0")
    P = None
    if args.precond:      # one KPM expansion per chain (its own Ē, spectrum bounds, orders and coefficients)
        P = pc.SymmetricKPMPreconditioner(m, 20, 0.05, 1.0, 1.0)
        if nchains > 1:
            pc.setup_chains_(P, rng=np.random.default_rng(7 + rank))
        else:
            pc.setup_(P, rng=np.random.default_rng(7 + rank))
    Bc = np.ascontiguousarray(B)

    def run(what_, nrhs_, reps, graph=0):
        ms = C.c_double()
        check(lib.elph_bench_run(m._h, what_, nrhs_, reps, graph, C.byref(ms)))
        return ms.value

    K = max(1, args.steps)                   # exactly the K and W asked for (every step is its own pair of launches)
    W = max(0, args.warmup)

    prep = 1 if what == 9 else what
    check(lib.elph_bench_prepare(m._h, prep, nrhs, _lib.dptr(Bc)))
    if two_streams:
        try:
            run(11, nrhs, 2)
            what = 11
        except _lib.ElphError:      # halves that are not whole groups of chains, or no p/x-fused iteration for this shape: one stream
            two_streams = False
            check(lib.elph_bench_prepare(m._h, prep, nrhs, None))
    if W:
        run(what, nrhs, W)                   # warm-up
    check(lib.elph_bench_prepare(m._h, prep, nrhs, None))
    ev = {}

    def run_steps(k):
        ev["ms"] = run(what, nrhs, k)        # exactly k steps; returns after the stream has drained
        check(lib.elph_synchronize(m._h))
        return 2.0 * nrhs * k

    check(lib.elph_synchronize(m._h))
    elapsed, matvecs = edist.timed_steps(comm, run_steps, K)
    ms_events = ev["ms"]
    fz_timed = C.c_int(0)                    # which form of the preconditioned iteration the timed region ran (0 unfused, 1 / 2 p/x-fused: lane program / registers)
    if args.precond:
        check(lib.elph_bench_px_info(m._h, C.byref(fz_timed)))

    out = None
    if rank == 0:
        ndim = m.Ndim
        out = {
            "metric": "cg_matvecs_per_sec",
            "value": matvecs / elapsed,
            "unit": "matvec/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": 1e3 * elapsed / K,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"BASELINE config {args.config}: {DESCR.get(args.config, args.config)} "
                            f"(N={m.Nsites}, Ltau={m.Ltau}, Ndim={m.Ndim}), "
                            f"{'KPM-preconditioned' if args.precond else 'un-preconditioned'} CG iteration, "
                            f"{nrhs} right-hand sides per step on each GPU = {nchains} independent Markov chain(s) x "
                            f"{nrhs // nchains} solve(s) each (2 = the two pseudofermion solves of one HMC force evaluation), "
                            f"{world} GPU(s) with their own chains",
                "nrhs": nrhs, "chains_per_gpu": nchains, "ndim": ndim, "preconditioned": bool(args.precond),
                "form": ("workgroup-resident: K iterations of the whole batch in one launch (k_cg_wg)" if resident else
                         "two-kernel iteration, one pair of launches per step (k_cg_ap + k_cg_xr%s)" % (" + KPM apply" if args.precond else "") +
                         (", as two half-batches on two streams" if two_streams else "")),
                "parallelism": f"gpus{world}xchains{nchains}",
            },
            "cg_iters_per_sec": nrhs * K * world / elapsed,
            "cg_batch_steps_per_sec": K / elapsed,
            "ms_per_step_events": ms_events / K,
            "build_info": lib.elph_build_info().decode(),
        }

        # ---- roofline ------------------------------------------------------------------------------------------------------------
        # (1) the two-kernel STREAMING iteration at the same batch: k_cg_ap and k_cg_xr timed alone with HIP events on their stream,
        #     priced on the compulsory bytes of each kernel as built against the HBM peak
        reps = 2000 if not resident else 400
        check(lib.elph_bench_prepare(m._h, 1, nrhs, None))
        run(4, nrhs, 320, graph=0)
        ms_ap = run(4, nrhs, reps, graph=0) / reps
        Tsl = C.c_int()
        check(lib.elph_bench_info(m._h, nrhs, C.byref(Tsl)))
        vec = 8.0 * ndim * nrhs                                   # one solver vector of the batch, bytes
        if m.kind == 1:   # SSH: E = exp(dtau mu) per site (negligible); the per-slice cosh/sinh tables once per chain
            tab = 16.0 * m.Ltau * m.Nbonds * nchains
            mv = 16.0 * ndim + 16.0 * m.Ltau * m.Nbonds           # SURVEY §8d mat-vec
            alg_ap, alg_it = (2 * mv + 48.0 * ndim) * nrhs, (2 * mv + 72.0 * ndim) * nrhs
        else:             # Holstein: E = exp(-dtau V) per (site, tau), once per chain
            tab = 8.0 * ndim * nchains
            alg_ap, alg_it = ALG_BYTES_PER_ELT["k_cg_ap"] * ndim * nrhs, ALG_BYTES_PER_ELT["cg_iter"] * ndim * nrhs
        built_ap = 6.0 * vec + tab                                # BUILT_BYTES_PER_ELT["k_cg_ap"] x elements + tables
        built_xr = 3.0 * vec
        ach = built_ap / (ms_ap * 1e-3) / 1e9
        kname = f"k_cg_ap_chunk<T={Tsl.value}>" if Tsl.value > 1 else "k_cg_ap_fast"
        tkey = f"{kname}|config={args.config}|nrhs={nrhs}|chains={nchains}"
        tpath = os.path.join(ROOT, "profiles", "traffic.json")

        def pmc(key, field):      # PMC bytes are only valid for the exact kernel / batch / shape they were collected on
            try:
                ent = json.load(open(tpath)).get("kernels", {}).get(key)
                return ent[field] if ent is not None else None
            except Exception:
                return None

        traffic = pmc(tkey, "hbm_bytes_per_launch")
        check(lib.elph_bench_prepare(m._h, 1, nrhs, None))
        run(5, nrhs, 320, graph=0)
        ms_xr = run(5, nrhs, reps, graph=0) / reps
        if resident:
            check(lib.elph_bench_prepare(m._h, 1, nrhs, None))
            run(1, nrhs, 40)
            ms_stream = run(1, nrhs, 400) / 400
        else:
            ms_stream = ms_events / K
        stream = {
            "bound": "hbm", "kernel": kname, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
            "traffic": traffic, "traffic_key": tkey, "avg_launch_us": ms_ap * 1e3, "bytes_per_launch": built_ap,
            "bytes_model": "compulsory bytes of k_cg_ap as built: 6 vectors (reads r|P^-1 r, p_old, x; writes p, z, x) x 8 B x Ndim x nrhs "
                           "+ exp(-dtau V) (SSH: per-slice hopping tables) once per chain",
            "algorithmic_GBs": alg_ap / (ms_ap * 1e-3) / 1e9,          # SURVEY 8(d)'s UNFUSED 96 B/elt: not a roofline fraction
            "xr_kernel_us": ms_xr * 1e3, "xr_bytes_per_launch": built_xr, "xr_frac": built_xr / (ms_xr * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "iteration_us": ms_stream * 1e3, "iteration_bytes": built_ap + built_xr,
            "iteration_frac": (built_ap + built_xr) / (ms_stream * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "iteration_matvecs_per_sec": 2.0 * nrhs / (ms_stream * 1e-3),
        }
        if traffic:      # what the memory-side counters saw per launch
            stream["traffic_frac"] = traffic / (ms_ap * 1e-3) / 1e9 / HBM_PEAK_GBS
            stream["traffic_over_bytes"] = traffic / built_ap
        if not resident and args.precond and m.kind == 0:
            # --precond: the timed region IS the preconditioned iteration (k_cg_ap + forward transform with the residual update + Chebyshev +
            # inverse transform with the p/x-update; two half-batches on two streams from 192 right-hand sides): priced on the compulsory
            # bytes of all its kernels as built against the HBM peak, and against the streaming rate a read-read-write mix reaches on this
            # part once it is larger than the Infinity Cache (tools/probes/access_pattern_probe.cpp, profiles/r06/)
            fz = fz_timed                    # (asked right after the timed region: the legs in between have re-planned the handle)
            it_bytes = ((2.0 * vec + tab) + 4.0 * vec + 2.0 * vec + 5.0 * vec) if fz.value else (built_ap + 4.0 * vec + 2.0 * vec + 2.0 * vec)
            us_it = 1e3 * ms_events / K
            achp = it_bytes / (us_it * 1e-6) / 1e9
            out["roofline"] = {
                "bound": "hbm", "kernel": ("k_cg_ap_sq16_px" if fz.value == 2 else ("k_cg_ap_chunk_px" if fz.value else kname)) +
                                          " + k_dft_mfma_r2s(forward, residual update) + k_kpm_cheb_sq + k_dft_mfma_r2s(inverse, p/x update)",
                "achieved": achp, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achp / HBM_PEAK_GBS, "traffic": None,
                "avg_launch_us": us_it, "bytes_per_launch": it_bytes, "px_fused": int(fz.value), "two_streams": bool(two_streams),
                "hbm_streaming_ceiling_GBs": HBM_STREAM_CEILING_GBS, "frac_of_streaming_ceiling": achp / HBM_STREAM_CEILING_GBS,
                "bytes_model": "p/x-fused: k_cg_ap reads p (+ tables), writes z (2 vectors + tables); forward transform reads r, z, writes r, nu (4); "
                               "Chebyshev reads and writes nu (2); inverse reads nu, p, x, writes p, x (5) — 13 vectors x 8 B x Ndim x nrhs",
            }
            out["roofline_streaming_unpreconditioned"] = stream
        elif not resident:
            out["roofline"] = stream
        else:
            # (2) the RESIDENT kernel — the timed region above is ONE launch of k_cg_wg (K iterations of all right-hand sides).  Its
            # Krylov vectors never leave the chip: per launch it reads r0 (= p0), x0, exp(-dtau V) once and writes x once, so HBM does not
            # bind it (hbm_frac is small BY DESIGN).  What binds it is f64 vector issue (v_fma_f64 + DPP moves of the checkerboard) and
            # the ONE team meeting per iteration through L2 — so the roofline it is held against is the f64 vector peak:
            #   flops per right-hand side and iteration = 2 mat-vecs x (2 Ndim + 6 Ltau Nbonds) + 10 Ndim (p.z, x, r, r.r, p)  [SURVEY 8(d)]
            us_launch = ms_events * 1e3
            flops = float(K) * nrhs * (2.0 * (2.0 * m.Ndim + 6.0 * m.Ltau * m.Nbonds) + 10.0 * m.Ndim)
            tfl = flops / (us_launch * 1e-6) / 1e12
            built_wg = 4.0 * vec + tab                        # reads r0 (also p0), x0 + tables; writes x (and r at the end of a sharded solve)
            hbm_GBs = built_wg / (us_launch * 1e-6) / 1e9
            wkey = f"k_cg_wg<T={wg_T.value},W={wg_W.value},G={wg_G.value}>|config={args.config}|nrhs={nrhs}|chains={nchains}"
            per_it = pmc(wkey, "hbm_bytes_per_iteration")
            sync_share = None
            try:      # share of an iteration a wave spends in the meeting, from the stamped diagnostic build (profiles/, not measured live)
                sync_share = json.load(open(tpath)).get("wg_sync_share", {}).get(f"T={wg_T.value}")
            except Exception:
                pass
            out["roofline"] = {
                "bound": "f64_vector+sync", "kernel": f"k_cg_wg<T={wg_T.value},W={wg_W.value},G={wg_G.value}>",
                "achieved": tfl, "peak": F64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tfl / F64_MFMA_PEAK_TFLOPS,
                "traffic": (per_it * K) if per_it else None, "traffic_key": wkey,
                "flops_per_launch": flops, "avg_launch_us": us_launch, "iterations_per_launch": K, "us_per_iteration": us_launch / K,
                "f64_TFLOPs": tfl, "f64_frac": tfl / F64_MFMA_PEAK_TFLOPS,
                "sync_share_profiled": sync_share,
                "meetings_per_iteration": 1,
                "hbm_bytes_per_launch": built_wg, "hbm_GBs": hbm_GBs, "hbm_frac": hbm_GBs / HBM_PEAK_GBS,
                "hbm_streaming_equivalent_GBs": alg_it * K / (us_launch * 1e-6) / 1e9,   # what a streaming implementation of 8(d)'s 120 B x Ndim would need
                "slices_per_wave": wg_T.value, "waves_per_workgroup": wg_W.value, "workgroups_per_rhs": wg_G.value,
                "workgroups_in_grid": 8 * ((nrhs + 7) // 8) * wg_G.value,
                "streaming_ap_kernel": kname, "streaming_ap_us": stream["avg_launch_us"], "streaming_ap_frac": stream["frac"],
                "streaming_ap_traffic": traffic, "streaming_xr_us": stream["xr_kernel_us"], "streaming_xr_frac": stream["xr_frac"],
                "streaming_iter_us": stream["iteration_us"], "streaming_iter_frac": stream["iteration_frac"],
                "streaming_matvecs_per_sec": stream["iteration_matvecs_per_sec"],
                "speedup_over_streaming": stream["iteration_us"] / (us_launch / K),
                "note": "flops per SURVEY 8(d); peak = f64 vector (= f64 MFMA) 78.6 TFLOP/s; hbm_frac is small by design (vectors stay on chip)",
            }
            out["roofline_streaming"] = stream

        # ---- the preconditioned iteration (BASELINE config C "with tau-FFT FourierAcceleration precond"): every kernel of it
        # timed alone with HIP events, priced on the compulsory bytes of ALL its kernels, and the matrix-core rate of the two tau-transforms
        if not args.no_sweep and m.kind == 0:
            try:
                Pr = P or pc.SymmetricKPMPreconditioner(m, 20, 0.05, 1.0, 1.0)
                (pc.setup_chains_ if nchains > 1 else pc.setup_)(Pr, rng=np.random.default_rng(7))
                rp = {}
                names = {4: "ap", 6: "fwd", 7: "cheb", 8: "inv", 3: "iter"}
                Hh, Qq = m.Ltau // 2, (m.Ltau // 2 + 1) // 2
                gemm_flops = 2.0 * (2.0 * (2 * Qq) * Hh) * m.Nsites * nrhs if m.Ltau % 2 == 0 else None   # two half-length f64 GEMMs (even / odd slices)
                # k_cg_ap 6 vectors + tables; forward transform with the residual update folded in: reads r, z, writes r, nu (4; the half
                # spectrum nu is one vector of bytes); Chebyshev reads and writes nu (2); inverse reads nu, writes P^-1 r (2)
                check(lib.elph_bench_prepare(m._h, 3, nrhs, None))
                run(3, nrhs, 2)                               # (which k_cg_ap form runs is known once one has run)
                fused = C.c_int()
                check(lib.elph_bench_px_info(m._h, C.byref(fused)))
                if fused.value:      # p/x-fused (round 5): k_cg_ap reads the ready p (+ tables), writes z; the inverse transform reads nu, p, x and
                    byts = {4: 2.0 * vec + tab, 6: 4.0 * vec, 7: 2.0 * vec, 8: 5.0 * vec}      # writes p, x: 13 vectors per iteration instead of 14
                else:
                    byts = {4: built_ap, 6: 4.0 * vec, 7: 2.0 * vec, 8: 2.0 * vec}
                rp["precond_px_fused"] = int(fused.value)
                byts[3] = byts[4] + byts[6] + byts[7] + byts[8]
                for wh in (4, 6, 7, 8, 3):
                    check(lib.elph_bench_prepare(m._h, 3, nrhs, None))
                    run(3, nrhs, 2)                           # sane p.z partials / states for the kernels timed alone
                    run(wh, nrhs, 32)
                    us = 1e3 * run(wh, nrhs, 320) / 320
                    rp[f"precond_{names[wh]}_us"] = us
                    rp[f"precond_{names[wh]}_hbm_frac"] = byts[wh] / (us * 1e-6) / 1e9 / HBM_PEAK_GBS
                    if wh in (6, 8) and gemm_flops:
                        rp[f"precond_{names[wh]}_mfma_frac"] = gemm_flops / (us * 1e-6) / 1e12 / F64_MFMA_PEAK_TFLOPS
                # the form elph_ldiv_batched runs from 192 right-hand sides: the same iteration as two half-batches on two streams
                try:
                    check(lib.elph_bench_prepare(m._h, 3, nrhs, None))
                    run(11, nrhs, 32)
                    rp["precond_iter_us_one_stream"] = rp["precond_iter_us"]
                    us2 = 1e3 * run(11, nrhs, 320) / 320
                    rp["precond_iter_us_two_streams"] = us2
                    if nrhs >= split_from and os.environ.get("ELPH_SPLIT_STREAMS") != "0":       # what a solve of this batch runs
                        rp["precond_iter_us"] = us2
                        rp["precond_iter_hbm_frac"] = byts[3] / (us2 * 1e-6) / 1e9 / HBM_PEAK_GBS
                except Exception as e:
                    rp["precond_two_streams_error"] = repr(e)
                rp["precond_bytes"] = byts[3]
                rp["precond_hbm_frac"] = rp["precond_iter_hbm_frac"]
                rp["precond_hbm_streaming_ceiling_GBs"] = HBM_STREAM_CEILING_GBS      # measured: read-read-write beyond the Infinity Cache (profiles/r06)
                rp["precond_frac_of_streaming_ceiling"] = byts[3] / (rp["precond_iter_us"] * 1e-6) / 1e9 / HBM_STREAM_CEILING_GBS
                rp["precond_ap_kernel"] = {0: "k_cg_ap_chunk (unfused)", 1: "k_cg_ap_chunk_px (lane program)", 2: "k_cg_ap_sq16_px (checkerboard in registers)"}[int(fused.value)]
                rp["precond_matvecs_per_sec"] = 2.0 * nrhs / (rp["precond_iter_us"] * 1e-6)
                # co-headline: the PRODUCTION path (every deck has [solver.preconditioner]; BASELINE config 3 "with tau-FFT precond") — the
                # same batch, one KPM-preconditioned CG iteration per step (k_cg_ap + forward transform with the residual update +
                # Chebyshev + inverse transform), timed in flight with HIP events
                out["precond_value"] = rp["precond_matvecs_per_sec"]
                out["precond_unit"] = "matvec/s"
                out["precond_ms_per_step"] = rp["precond_iter_us"] * 1e-3
                out["precond_cg_iters_per_sec"] = nrhs / (rp["precond_iter_us"] * 1e-6)
                out["roofline"].update(rp)
                out["roofline_preconditioned"] = dict(rp, note=(
                    "bytes (unfused): k_cg_ap 6 vectors + tables (src = P^-1 r); forward transform reads r, z and writes r, nu (4 vectors); Chebyshev reads "
                    "and writes nu; inverse reads nu, writes P^-1 r — 14 vectors per iteration.  p/x-fused (precond_px_fused = 1): k_cg_ap reads p, writes z "
                    "(2 + tables); the inverse reads nu, p, x and writes p, x (5) — 13.  SURVEY 8(d)'s 16 B x Ndim per KPM apply "
                    "assumes the three kernels fused into one.  The Chebyshev kernel is latency-bound by its longest recursion.  mfma: two "
                    f"(L/2 x L/2) real f64 GEMMs per transform on v_mfma_f64_16x16x4_f64, peak {F64_MFMA_PEAK_TFLOPS} TFLOP/s"))
            except Exception as e:
                out["roofline_preconditioned"] = {"error": repr(e)}

        # ---- secondary: time to solution of the whole default batch (every chain its own matrix), plain vs KPM
        if not args.no_sweep and m.kind == 0:
            try:
                Pb = P or pc.SymmetricKPMPreconditioner(m, 20, 0.05, 1.0, 1.0)
                setup = pc.setup_chains_ if nchains > 1 else pc.setup_
                setup(Pb, rng=np.random.default_rng(7))          # first call allocates the per-chain tables
                tk = time.perf_counter()
                setup(Pb, rng=np.random.default_rng(7))
                t_setup = time.perf_counter() - tk
                bt = {"kpm_setup_ms_all_chains": 1e3 * t_setup}
                for label, PP in (("plain", None), ("kpm", Pb)):
                    X = np.zeros_like(Bc)
                    models.ldiv_batched_(X, m, Bc, P=PP)
                    X[:] = 0.0
                    tq = time.perf_counter()
                    it, res, fl = models.ldiv_batched_(X, m, Bc, P=PP)
                    dtq = time.perf_counter() - tq
                    bt[label] = {"ms_per_batched_solve_incl_pcie": 1e3 * dtq, "solves_per_sec": nrhs / dtq,
                                 "iters_max": int(it.max()), "iters_mean": float(it.mean()),
                                 "flags_ok": bool((fl == 0).all()), "max_residual": float(res.max())}
                ms = C.c_double()
                for wh, nm in ((2, "kpm_apply_us"), (3, "preconditioned_cg_iter_us")):
                    check(lib.elph_bench_prepare(m._h, wh, nrhs, None))
                    check(lib.elph_bench_run(m._h, wh, nrhs, 32, 0, C.byref(ms)))
                    check(lib.elph_bench_run(m._h, wh, nrhs, 320, 0, C.byref(ms)))
                    bt[nm] = 1e3 * ms.value / 320
                out[f"batch_time_to_solution_tol1e-5_nrhs{nrhs}_chains{nchains}"] = bt
            except Exception as e:
                out[f"batch_time_to_solution_tol1e-5_nrhs{nrhs}_chains{nchains}"] = {"error": str(e)}
        if nchains > 1:
            models.update_model_(m)          # back to one configuration for the secondary single-matrix measurements
            if args.precond:                 # (its expansion went with the chains' matrices)
                pc.setup_(P, rng=np.random.default_rng(7 + rank))
        # ---- secondary: the resident kernel on the other BASELINE lattices (configs B: square L = 8, Ntau = 40; D: honeycomb L = 12,
        # Ntau = 120; E: optical SSH L = 16, Ntau = 160 — parity-test cases, SURVEY §8 sizes table), 256 right-hand sides,
        # un-preconditioned iteration
        if not args.no_sweep and rank == 0 and args.config == "C" and resident:
            other = {}
            for tag in ("B", "D", "E", "T"):      # (T: the triangular deck's geometry at config C's size — the resident form of round 4)
                try:
                    mo_ = configs.make_model(tag, tol=1e-5, device=comm.device_index())
                    _, Bo = configs.rhs(mo_, 256)
                    msd = C.c_double()
                    for reps in (160, 1600):
                        check(lib.elph_bench_prepare(mo_._h, 1, 256, _lib.dptr(np.ascontiguousarray(Bo))))
                        check(lib.elph_bench_run(mo_._h, 9, 256, reps, 0, C.byref(msd)))
                    other[tag] = {"us_per_step": 1e3 * msd.value / 1600, "matvecs_per_sec": 2.0 * 256 * 1600 / (msd.value * 1e-3), "nrhs": 256}
                    out["roofline"][f"config_{tag}_matvecs_per_sec_nrhs256"] = other[tag]["matvecs_per_sec"]
                    mo_.close()
                except Exception as e:
                    other[tag] = {"error": str(e)}
            out["other_baseline_lattices_resident"] = other
            # ---- secondary: lattices of production size beyond the BASELINE ones (square 32 x 32, honeycomb 24 x 24 cells, triangular 24 x 24: the PGRID
            # kernels, a patch of sites per lane) — the streaming iterations, un-preconditioned and KPM-preconditioned, 72 right-hand sides
            large = {}
            for tag in ("X32", "X24", "XT24"):
                try:
                    ml_ = configs.make_model(tag, tol=1e-5, device=comm.device_index())
                    Pl_ = pc.SymmetricKPMPreconditioner(ml_, 20, 0.05, 1.0, 1.0)
                    pc.setup_(Pl_, rng=np.random.default_rng(7))
                    Bl = np.ascontiguousarray(np.stack([synth.randn(300 + r, ml_.Ndim) for r in range(72)]))
                    rec = {"nsites": int(ml_.Nsites), "ltau": int(ml_.Ltau), "nrhs": 72}
                    msl = C.c_double()
                    for wh, nm in ((1, "cg_iter_us"), (3, "preconditioned_cg_iter_us")):
                        check(lib.elph_bench_prepare(ml_._h, wh, 72, _lib.dptr(Bl)))
                        check(lib.elph_bench_run(ml_._h, wh, 72, 32, 0, C.byref(msl)))
                        check(lib.elph_bench_prepare(ml_._h, wh, 72, None))
                        check(lib.elph_bench_run(ml_._h, wh, 72, 160, 0, C.byref(msl)))
                        rec[nm] = 1e3 * msl.value / 160
                    # (round 6) the preconditioned iteration as a solve of this batch runs it: p/x-fused, two half-batches on two streams
                    try:
                        fz_ = C.c_int()
                        check(lib.elph_bench_prepare(ml_._h, 3, 72, None))
                        check(lib.elph_bench_px_info(ml_._h, C.byref(fz_)))
                        rec["preconditioned_px_fused"] = int(fz_.value)
                        check(lib.elph_bench_run(ml_._h, 11, 72, 32, 0, C.byref(msl)))
                        check(lib.elph_bench_run(ml_._h, 11, 72, 160, 0, C.byref(msl)))
                        rec["preconditioned_cg_iter_us_one_stream"] = rec["preconditioned_cg_iter_us"]
                        rec["preconditioned_cg_iter_us"] = rec["preconditioned_cg_iter_us_two_streams"] = 1e3 * msl.value / 160
                    except _lib.ElphError:
                        pass                             # (no two-stream form for this shape: the one-stream figure stands)
                    # one right-hand side (the reference's call shape): the streaming pair and — where the rule of slabs.hip takes it —
                    # the slab form (the resident kernel on slabs of rows of the lattice, all slabs one launch)
                    for wh, nm in ((1, "cg_iter_us_1rhs_streaming"), (12, "cg_iter_us_1rhs_slabs")):
                        try:
                            for reps in (64, 400):
                                check(lib.elph_bench_prepare(ml_._h, 1, 1, _lib.dptr(np.ascontiguousarray(Bl[:1]))))
                                check(lib.elph_bench_run(ml_._h, wh, 1, reps, 0, C.byref(msl)))
                            rec[nm] = 1e3 * msl.value / 400
                        except _lib.ElphError:
                            rec[nm] = None               # (the slab form does not apply to this lattice)
                    rec["preconditioned_ps_per_element"] = 1e6 * rec["preconditioned_cg_iter_us"] / (72.0 * ml_.Ndim)
                    rec["preconditioned_matvecs_per_sec"] = 2.0 * 72 / (rec["preconditioned_cg_iter_us"] * 1e-6)
                    large[tag] = rec
                    ml_.close()
                except Exception as e:
                    large[tag] = {"error": str(e)}
            out["large_lattices"] = large
        # ---- secondary: the same step at other batch sizes (short runs)
        if not args.no_sweep:
            sweep = {}
            for nr in (1, 2, 10, 24, 48, 64, 128, 256):
                _, Bs = configs.rhs(m, nr)
                row = {}
                for form, wh in ((("resident", 9),) if resident else ()) + (("streaming", 3 if args.precond else 1),):
                    pw = 1 if wh == 9 else wh
                    check(lib.elph_bench_prepare(m._h, pw, nr, _lib.dptr(np.ascontiguousarray(Bs))))
                    run(wh, nr, 160)
                    check(lib.elph_bench_prepare(m._h, pw, nr, None))
                    ms = run(wh, nr, 1600)
                    row[form] = {"us_per_step": 1e3 * ms / 1600, "matvecs_per_sec": 2.0 * nr * 1600 / (ms * 1e-3)}
                best = max(row.values(), key=lambda e: e["matvecs_per_sec"])
                sweep[str(nr)] = {"us_per_step": best["us_per_step"], "matvecs_per_sec": best["matvecs_per_sec"],
                                  "alg_GBs": ALG_BYTES_PER_ELT["cg_iter"] * ndim * best["matvecs_per_sec"] / 2.0 / 1e9, **row}
            out["by_nrhs"] = sweep

        # ---- secondary: time-to-solution of one ldiv! at tol=1e-5, plain vs KPM-preconditioned (BASELINE config 3)
        if not args.no_sweep:
            try:
                P2 = P or pc.SymmetricKPMPreconditioner(m, 20, 0.05, 1.0, 1.0)
                pc.setup_(P2, rng=np.random.default_rng(7))
                tts = {}
                for nr in (1, 10):
                    _, Bs = configs.rhs(m, nr)
                    for label, PP in (("plain", None), ("kpm", P2)):
                        X = np.zeros_like(Bs)
                        models.ldiv_batched_(X, m, Bs, P=PP)                    # warm
                        X[:] = 0.0
                        tq = time.perf_counter()
                        it, res, fl = models.ldiv_batched_(X, m, Bs, P=PP)
                        dtq = time.perf_counter() - tq
                        tts[f"{label}_nrhs{nr}"] = {"ms_per_batched_solve_incl_pcie": 1e3 * dtq, "iters_max": int(it.max()),
                                                    "flags_ok": bool((fl == 0).all()), "max_residual": float(res.max())}
                ms = C.c_double()
                check(lib.elph_bench_prepare(m._h, 2, 1, None))
                check(lib.elph_bench_run(m._h, 2, 1, 160, 0, C.byref(ms)))
                tts["kpm_apply_us_nrhs1"] = 1e3 * ms.value / 160
                check(lib.elph_bench_prepare(m._h, 3, 1, None))
                check(lib.elph_bench_run(m._h, 3, 1, 160, 0, C.byref(ms)))
                tts["preconditioned_cg_iter_us_nrhs1"] = 1e3 * ms.value / 160
                tts["kpm_orders_sum"] = int(P2.orders.sum())
                out["time_to_solution_tol1e-5"] = tts
                out["roofline"]["precond_iter_us_nrhs1"] = tts["preconditioned_cg_iter_us_nrhs1"]
                out["roofline"]["kpm_apply_us_nrhs1"] = tts["kpm_apply_us_nrhs1"]
            except Exception as e:
                out["time_to_solution_tol1e-5"] = {"error": str(e)}

        # ---- secondary: whole HMC updates (the caller of the path): chains in lockstep on the GPU vs the CPU
        # oracle's single chain (the reference's configuration), KPM-preconditioned, same deck
        if not args.no_sweep and m.kind == 0:
            try:
                from elphdynamics_amd import hmc as ehmc
                nt_g, dt_h = 10, 0.01
                hm = {}
                for nch_h in (1, HMC_CHAINS):
                    mh = configs.make_model(args.config, tol=1e-5, maxiter=20000, device=comm.device_index())
                    fah = pc.FourierAccelerator(mh)
                    pc.update_M_(fah, mh, 0.0, np.inf, 1.0, 0.1)
                    Hh = ehmc.HybridMonteCarlo(mh, fah, dt=dt_h, tr=nt_g * dt_h, alpha=0.0, Nb=1, nchains=nch_h)
                    if nch_h > 1:
                        Hh.X[:] = np.stack([synth.phonon_field(mh.Nph, mh.Ltau, mh.beta, mh.dtau, seed=100 + 17 * c) for c in range(nch_h)])
                        Hh.push_()
                    Ph = pc.SymmetricKPMPreconditioner(mh, 20, 0.05, 1.0, 1.0)
                    Hh.device_rng_(3)      # random inputs drawn inside the library (on the GPU), as a production run would
                    upd = (lambda: ehmc.update_chains_(mh, Hh, fah, Ph)) if nch_h > 1 else (lambda: ehmc.update_(mh, Hh, fah, Ph, pull=False))
                    upd()
                    # (1) the update right after the warm-up one, on the synthetic start field: the number rounds 2 and 3 reported
                    tq = time.perf_counter()
                    acc_h, its_h = upd()
                    dth = time.perf_counter() - tq
                    rec = {"nt": nt_g, "ms_per_update": 1e3 * dth, "ms_per_chain_update": 1e3 * dth / nch_h,
                           "chain_evaluations_per_sec": nch_h * (nt_g + 2) / dth,
                           "iters_per_solve": float(np.mean(its_h)), "accepted": float(np.mean(acc_h)),
                           "random_numbers": "library generator (elph_hmc_set_rng)"}
                    # (2) steady state of the production caller: updates back to back for ~hmc_seconds — the field leaves its cold start
                    # and the solves need more iterations (30 -> ~50 at config C), so this is NOT comparable with (1) per update, only
                    # per iteration; also the one stretch of the run long enough for a 5 s GPU-activity sampler to see
                    if nch_h > 1 and args.hmc_seconds > 0:
                        n_upd, acc_all, its_all = 0, [], []
                        tq = time.perf_counter()
                        while True:
                            acc_h, its_h = upd()
                            n_upd += 1
                            acc_all.append(np.mean(acc_h)); its_all.append(np.mean(its_h))
                            dts = time.perf_counter() - tq
                            if dts >= args.hmc_seconds:
                                break
                        per = dts / n_upd
                        rec["steady_state"] = {"updates_timed": n_upd, "seconds": dts, "ms_per_update": 1e3 * per, "ms_per_chain_update": 1e3 * per / nch_h,
                                               "iters_per_solve": float(np.mean(its_all)), "accepted": float(np.mean(acc_all)),
                                               "us_per_chain_update_and_cg_iteration": 1e6 * per / nch_h / float(np.mean(its_all))}
                    hm[f"gpu_chains{nch_h}"] = rec
                    mh.close()
                if not args.no_cpu:
                    from oracle.oracle import Oracle
                    orc_h = Oracle(fast=True)
                    mo = configs.make_model(args.config, tol=1e-5, maxiter=20000, device=comm.device_index())
                    fao = pc.FourierAccelerator(mo)
                    pc.update_M_(fao, mo, 0.0, np.inf, 1.0, 0.1)
                    Eo = orc_h.update_model_holstein(mo.Nsites, mo.Ltau, mo.dtau, mo.x, mo.lam, mo.lam2, mo.mu)
                    omo = orc_h.make_model(0, mo.Nsites, mo.Ltau, mo.neighbor_table, mo.cosht, mo.sinht, Eo)
                    Po = orc_h.make_kpm(omo, n=20)
                    nt_c = 1
                    rngc = np.random.default_rng(3)
                    rnd = dict(R=rngc.standard_normal(mo.Ndof), Rp=rngc.standard_normal(mo.Ndim), Rm=rngc.standard_normal(mo.Ndim),
                               kpm_randn=rngc.standard_normal((nt_c + 2) * 2 * mo.Nsites), u=0.5)
                    tq = time.perf_counter()
                    orc_h.hmc_update_holstein(omo, mo.x.copy(), np.zeros(mo.Ndof), mo.omega, mo.omega4, mo.lam, mo.lam2, mo.mu, mo.dtau,
                                              fao.M, dt_h, nt_c, 1, 0.0, rnd, P=Po, tol=1e-5, maxiter=20000)
                    dtc = time.perf_counter() - tq
                    hm["cpu_oracle_1core"] = {"nt": nt_c, "s_per_update": dtc, "chain_evaluations_per_sec": (nt_c + 2) / dtc}
                    hm["gpu_over_cpu_chain_evaluations"] = hm[f"gpu_chains{HMC_CHAINS}"]["chain_evaluations_per_sec"] / hm["cpu_oracle_1core"]["chain_evaluations_per_sec"]
                    mo.close()
                out["hmc_update_kpm"] = hm
                out["roofline"]["hmc_update_ms_1chain"] = hm["gpu_chains1"]["ms_per_update"]
                out["roofline"][f"hmc_chain_update_ms_{HMC_CHAINS}chains"] = hm[f"gpu_chains{HMC_CHAINS}"]["ms_per_chain_update"]
                ss = hm[f"gpu_chains{HMC_CHAINS}"].get("steady_state")
                if ss:      # thermalised field: more CG iterations per solve than on the synthetic start — compare per iteration
                    out["roofline"][f"hmc_steady_chain_update_ms_{HMC_CHAINS}chains"] = ss["ms_per_chain_update"]
                    out["roofline"][f"hmc_steady_iters_per_solve_{HMC_CHAINS}chains"] = ss["iters_per_solve"]
            except Exception as e:
                out["hmc_update_kpm"] = {"error": repr(e)}

        # ---- CPU baseline: the oracle, 1 thread, bounded sample of the same workload
        if not args.no_cpu:
            try:
                from oracle.oracle import Oracle
                orc = Oracle(fast=True)
                if m.kind == 0:
                    E = orc.update_model_holstein(m.Nsites, m.Ltau, m.dtau, m.x, m.lam, m.lam2, m.mu)
                    om = orc.make_model(0, m.Nsites, m.Ltau, m.neighbor_table, m.cosht, m.sinht, E)
                else:
                    om = orc.make_model(1, m.Nsites, m.Ltau, m.neighbor_table, np.ascontiguousarray(m.cosht).reshape(-1),
                                        np.ascontiguousarray(m.sinht).reshape(-1), m.expDtauMu)
                b0 = np.ascontiguousarray(B[0])
                tt = time.perf_counter()
                orc.cg_solve(om, b0, tol=0.0, maxiter=200)
                per_it = (time.perf_counter() - tt) / 200
                n_it = int(max(200, min(200000, args.cpu_seconds / per_it)))
                tt = time.perf_counter()
                orc.cg_solve(om, b0, tol=0.0, maxiter=n_it)
                dt = time.perf_counter() - tt
                out["cpu_baseline"] = {"value": 2.0 * n_it / dt, "unit": "matvec/s", "cores": 1, "kind": "port",
                                       "sample": f"{n_it} un-preconditioned CG iterations (tol=0) on right-hand side 0 of the same "
                                                 f"config-{args.config} workload, oracle/elph_oracle.c built -O3 -march=native "
                                                 f"-ffast-math, single thread ({os.cpu_count()} host cores present)",
                                       "cg_iters_per_sec": n_it / dt, "seconds": dt}
                # all host cores THE REFERENCE'S WAY: it has no threads (ElPhDynamics.jl:74-75 pins BLAS and FFTW to one) and fills a node
                # with independent run-IDs, one single-threaded process per core (ElPhDynamics.jl:90-95) — so: one single-threaded oracle
                # solve per usable core, all at once, each on its own copy of the model; value = the sum of their rates.  (Round 4's
                # OpenMP variant of ONE solve did not scale — 1.15x on 8 threads — and was no baseline of anything: dropped.)
                try:
                    import subprocess
                    import tempfile
                    visible = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
                    cores = min(visible, int(os.environ.get("ELPH_BENCH_CPU_WORKERS", "16")))      # a one-GPU box's CPU share is 16 cores, whatever it shows
                    n_each = int(max(200, min(200000, 0.5 * args.cpu_seconds / per_it)))
                    with tempfile.TemporaryDirectory() as td:
                        f_in = os.path.join(td, "chain.npz")
                        if m.kind == 0:
                            np.savez(f_in, kind=0, N=m.Nsites, L=m.Ltau, table=m.neighbor_table, c=m.cosht, s=m.sinht, E=E, b=b0, n=n_each)
                        else:
                            np.savez(f_in, kind=1, N=m.Nsites, L=m.Ltau, table=m.neighbor_table, c=np.ascontiguousarray(m.cosht).reshape(-1),
                                     s=np.ascontiguousarray(m.sinht).reshape(-1), E=m.expDtauMu, b=b0, n=n_each)
                        # fresh CHILD processes (never an exec of this GPU-initialised one), CPU only: they import numpy and the oracle
                        child = ("import sys, time, numpy as np; sys.path.insert(0, %r); from oracle.oracle import Oracle; a = np.load(%r); "
                                 "o = Oracle(fast=True); om = o.make_model(int(a['kind']), int(a['N']), int(a['L']), a['table'], "
                                 "np.ascontiguousarray(a['c']), np.ascontiguousarray(a['s']), np.ascontiguousarray(a['E'])); b = np.ascontiguousarray(a['b']); "
                                 "o.cg_solve(om, b, tol=0.0, maxiter=50); t = time.perf_counter(); o.cg_solve(om, b, tol=0.0, maxiter=int(a['n'])); "
                                 "print(time.perf_counter() - t)") % (ROOT, f_in)
                        env = dict(os.environ, HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
                        procs = [subprocess.Popen([sys.executable, "-c", child], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, env=env)
                                 for _ in range(cores)]
                        secs = [float(pr.communicate(timeout=120 + 4 * args.cpu_seconds)[0].strip().splitlines()[-1]) for pr in procs]
                    out["cpu_baseline_all_cores"] = {
                        "value": sum(2.0 * n_each / t_ for t_ in secs), "unit": "matvec/s", "cores": cores, "kind": "port",
                        "per_core_min": 2.0 * n_each / max(secs), "per_core_max": 2.0 * n_each / min(secs),
                        "host_cores_visible": visible,
                        "sample": f"{cores} independent single-threaded processes at once ({n_each} iterations each), one per core of the box's CPU share — the "
                                  "reference's own way of using a node (independent run-IDs, ElPhDynamics.jl:90-95); value = sum of their rates"}
                except Exception as e:
                    out["cpu_baseline_all_cores"] = {"value": None, "note": f"failed: {e}"}
            except Exception as e:   # the baseline is a report, never a reason to lose the GPU number
                out["cpu_baseline"] = {"value": None, "unit": "matvec/s", "cores": 1, "kind": "port", "sample": f"failed: {e}"}
    m.close()
    # ---- the north_star's other curve: ONE solve of configs C, D, E sharded over the N ranks (all ranks take part; N = 1 included as
    # the first point of the 1 / 2 / 4 / 8 series unless the secondary measurements are off)
    spatial = None
    if not args.no_spatial and (world > 1 or not args.no_sweep):
        spatial = spatial_records(comm, args.spatial_steps, cpu=False)
    if rank == 0:
        if spatial is not None:
            out["spatial"] = spatial
            flat = {}
            for tag in ("C", "D", "E"):
                rec = spatial.get(tag) or {}
                if "us_per_iteration_device" in rec:
                    flat[f"spatial_{tag}_us_per_iteration"] = rec["us_per_iteration_device"]
                    flat[f"spatial_{tag}_matvecs_per_sec"] = rec["matvecs_per_sec"]
                    flat.setdefault("spatial_ranks", rec["ranks"])
                    flat.setdefault("rccl_ranks", rec["rccl_ranks"])
                    flat.setdefault("spatial_devices", len(rec.get("devices") or []))
            out.update(flat)
            out["roofline"].update(flat)
        out["roofline"] = order_roofline(out["roofline"])
        print(json.dumps(out))
    comm.close()


# The driver's record keeps the FIRST two dozen scalars of `roofline` (and drops every non-standard top-level key): what must survive
# goes first — the dominant kernel's line, then the production (KPM-preconditioned) iteration, the streaming form, the HMC update and
# the sharded solve.
ROOFLINE_FIRST = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_us", "us_per_iteration",
                  "precond_iter_us", "precond_hbm_frac", "precond_matvecs_per_sec", "precond_ap_us", "precond_fwd_us", "precond_cheb_us",
                  "precond_inv_us", "streaming_iter_us", "streaming_iter_frac", "hmc_update_ms_1chain",
                  "spatial_C_us_per_iteration", "spatial_C_matvecs_per_sec", "spatial_ranks", "rccl_ranks", "spatial_devices")


def order_roofline(roof):
    first = {k: roof[k] for k in ROOFLINE_FIRST if k in roof}
    first.update((k, v) for k, v in roof.items() if k not in first)
    return first


F64_FLOPS_PER_ITER = lambda ndim, ltau, nbonds: 2.0 * (2.0 * ndim + 6.0 * ltau * nbonds) + 10.0 * ndim     # noqa: E731  SURVEY §8(d)


TRANSPORT_TEXT = {
    "mailbox": "device-initiated stores into hipIpc / peer-mapped mailboxes, one inter-rank hop per iteration; no collective",
    "collectives": "torch.distributed collectives (nccl = RCCL over xGMI; gloo: staged through the host): two all-reduces (p.z, r.r) + one grouped "
                   "ghost-row send/recv per iteration, the library's streaming mat-vec on the slab (elphdynamics_amd/sharded_rccl.py)",
}


def spatial_record(tag, world, ranks_rccl, K, ms_dev_max, elapsed_max, N, Ltau, nbonds, slab, peer, selftest_us, backend, transport="mailbox", why=None):
    """The JSON sub-record of one sharded solve (pure arithmetic: the CPU test of the schema calls it at world 2 over gloo).
    ms_dev_max: MAX over ranks of the HIP-event time of the K-iteration launch; elapsed_max: MAX of the host time around it."""
    us_dev = 1e3 * ms_dev_max / K
    flops = F64_FLOPS_PER_ITER(N * Ltau, Ltau, nbonds)
    tfl = flops / (us_dev * 1e-6) / 1e12
    return {
        "config": tag, "workload": f"ONE un-preconditioned CG solve of BASELINE config {tag} (N={N}, Ltau={Ltau}) over {world} rank(s): {slab}",
        "ranks": world, "rccl_ranks": ranks_rccl, "dist_backend": backend, "iterations": K,
        "us_per_iteration_device": us_dev, "us_per_iteration_host": 1e6 * elapsed_max / K,
        "matvecs_per_sec": 2.0 / (us_dev * 1e-6), "cg_iters_per_sec": 1.0 / (us_dev * 1e-6),
        "bound": "f64_vector+sync", "achieved": tfl, "peak": F64_MFMA_PEAK_TFLOPS * world, "unit": "TFLOP/s",
        "frac": tfl / (F64_MFMA_PEAK_TFLOPS * world),
        "hbm_streaming_equivalent_frac": ALG_BYTES_PER_ELT["cg_iter"] * N * Ltau / (us_dev * 1e-6) / 1e9 / (HBM_PEAK_GBS * world),
        "peer_access": peer, "selftest_us_per_round": selftest_us,
        "scaling": "strong", "transport": transport, "transport_detail": TRANSPORT_TEXT.get(transport, transport), "transport_chosen_because": why,
    }


def peer_matrix(lib, ndev):
    """hipDeviceCanAccessPeer for every pair of the first ndev devices (None where the library cannot tell: no GPU)."""
    can = C.c_int()
    rows = []
    for a in range(ndev):
        row = []
        for b in range(ndev):
            row.append(int(can.value) if lib.elph_peer_access(a, b, C.byref(can)) == 0 else None)
        rows.append(row)
    return rows


def make_sharded(tag, comm):
    """ShardedSolver of a BASELINE config with its synthetic model set: (solver, b, N, Ltau, nbonds, slab description)."""
    import numpy as np
    from elphdynamics_amd import configs, lattice as lat, sharded, synth
    kind, norb, Ls, bonds, beta, dtau = configs.CONFIGS[tag]
    la = lat.Lattice(norb, Ls, Ls if Ls > 1 else 1, 1)
    raw = np.concatenate([la.calc_neighbor_table(o1, o2, d) for (o1, o2, d) in bonds], axis=0)
    cb = lat.initialize_checkerboard(raw, np.ones(raw.shape[0]), dtau)
    N, Ltau = la.nsites, lat.ltau_from_beta(beta, dtau)
    b = synth.rhs(N * Ltau)
    # ELPH_SHARD_TRANSPORT = mailbox | rccl | auto [auto: the in-library mailbox form when its preflight passes on every rank, else collectives]
    from elphdynamics_amd import sharded_rccl
    s, transport, why = sharded_rccl.make_solver(comm, norb, la.L1, la.L2, Ltau, cb["table"], cb["cosht"], cb["sinht"], kind=0 if kind == "holstein" else 1)
    s.bench_transport, s.bench_why = transport, why
    s.bench_geometry = (norb, la.L1, la.L2, Ltau, cb["table"], cb["cosht"], cb["sinht"], kind)
    if kind == "holstein":
        s.update_model(np.exp(-dtau * synth.phonon_field(N, Ltau, beta, dtau)))        # lambda = 1, mu = 0 (configs.py)
    else:
        nb = raw.shape[0]
        xb = 0.25 * synth.phonon_field(nb, Ltau, beta, dtau, omega=0.1, lam=0.0).reshape(nb, Ltau)
        tp = 1.0 - 0.1 * xb
        s.update_model_ssh(np.cosh(dtau * tp), np.sinh(dtau * tp), np.ones(N))
    slab = (f"slabs of rows of cells (+{s.sl['lo']}/{s.sl['hi']} ghost rows; {s.Nloc} of {N} sites on rank 0), " +
            ("resident CG kernel per rank, partial sums and boundary rows by device-initiated stores into mapped mailboxes" if transport == "mailbox"
             else "streaming mat-vec per rank, inner products by all-reduce, ghost rows by grouped send/recv"))
    s.bench_expV = np.exp(-dtau * synth.phonon_field(N, Ltau, beta, dtau)) if kind == "holstein" else None
    return s, b, N, Ltau, raw.shape[0], slab


def measure_sharded(tag, comm, K, W, factory=None):
    """One sharded solve of K iterations after W warm-up iterations: the sub-record (identical on every rank).  factory: stand-in for
    make_sharded (the CPU test of the record's schema; the product path has none)."""
    from elphdynamics_amd import _lib
    lib = _lib.load()
    s, b, N, Ltau, nb, slab = (factory or make_sharded)(tag, comm)
    if W:
        s.iterate(b, W)
    comm.barrier()
    t0 = time.perf_counter()
    ms_dev = s.iterate(b, K)                   # returns when this rank's launch has finished (includes prepare + barrier)
    comm.barrier()
    elapsed = comm.max(time.perf_counter() - t0)
    ms_dev = comm.max(ms_dev)
    rccl = None
    dist = getattr(comm, "dist", None) or getattr(getattr(getattr(comm, "sh", None), "proc_comm", None), "dist", None)
    if dist is not None and dist.is_initialized():
        rccl = int(dist.get_world_size())
    devs = sorted(set(comm.allgather_object(comm.device_index())))
    peer = peer_matrix(lib, max(devs) + 1) if lib.elph_device_count() > 0 else None
    st = [float(x) for x in s.selftest_us] if getattr(s, "selftest_us", None) is not None else None
    transport = getattr(s, "bench_transport", "mailbox")
    rec = spatial_record(tag, comm.world, rccl, K, ms_dev, elapsed, N, Ltau, nb, slab, peer, st, getattr(comm, "backend", None),
                         transport=transport, why=getattr(s, "bench_why", None))
    rec["devices"] = devs
    # A/B of the two transports on the same slabs (Holstein, more than one rank, a torch.distributed communicator): the collective form for a
    # tenth of the iterations — wall time per iteration (its host drives every iteration; the mailbox form's host only launches)
    geo = getattr(s, "bench_geometry", None)
    if transport == "mailbox" and comm.world > 1 and geo is not None and geo[7] == "holstein" and getattr(comm, "dist", None) is not None:
        s2, err = None, None
        try:
            from elphdynamics_amd import sharded_rccl
            s2 = sharded_rccl.CollectiveShardedSolver(comm, *geo[:7])
            s2.update_model(s.bench_expV)
        except Exception as e:      # noqa: BLE001 — agreed below: no rank enters the timed collectives alone
            err = repr(e)
        errs = comm.allgather_object(err)
        if not any(errs):
            k2 = max(16, K // 10)
            s2.iterate(b, 8)
            ms2 = comm.max(s2.iterate(b, k2))
            rec["ab_collectives"] = {"us_per_iteration_wall": 1e3 * ms2 / k2, "iterations": k2, "backend": getattr(comm, "backend", None),
                                     "collectives_per_iteration": 3, "note": TRANSPORT_TEXT["collectives"]}
        else:
            rec["ab_collectives"] = {"error": next(e for e in errs if e)}
        if s2 is not None:
            s2.close()
    s.close()
    return rec


def spatial_records(comm, K, cpu=False, factory=None):
    """Configs C, D and E over the ranks of `comm`; a config whose slabs do not fit this rank count is recorded with its reason."""
    out = {"note": "unmeasured across physical GPUs until a multi-GPU node runs this: on a one-GPU box every rank shares device 0 "
                   "(ELPH_FORCE_DEVICE) and the numbers are the protocol's cost, not xGMI's"} if len(set(comm.allgather_object(comm.device_index()))) < comm.world and comm.world > 1 else {}
    for tag in ("C", "D", "E"):
        out[tag] = measure_sharded_agreed(tag, comm, K, max(1, K // 10), factory)
    return out


def measure_sharded_agreed(tag, comm, K, W, factory=None):
    """measure_sharded with the ranks agreeing on failure BEFORE any of them enters a collective or a mailbox wait of the config: the
    set-up (slab geometry, handle, model, self-test) runs first and its outcome is all-gathered; if any rank failed, every rank skips
    the config and records the first error — a rank that failed alone (out of memory, a self-test error) cannot leave the others
    blocked in a barrier.  A failure inside the timed solve itself is bounded by the sharded solve's own wait bound
    (ELPH_SHARD_TIMEOUT_MS) and reported the same way."""
    made, err = None, None
    try:
        made = (factory or make_sharded)(tag, comm)
    except Exception as e:
        err = repr(e)
    errs = comm.allgather_object(err)
    if any(errs):
        if made is not None:
            try:
                made[0].close()
            except Exception:
                pass
        return {"config": tag, "ranks": comm.world, "error": next(e for e in errs if e), "failed_ranks": [r for r, e in enumerate(errs) if e]}
    rec, err = None, None
    try:
        rec = measure_sharded(tag, comm, K, W, factory=lambda *_: made)
    except Exception as e:
        err = repr(e)
    errs = comm.allgather_object(err)
    if any(errs):
        return {"config": tag, "ranks": comm.world, "error": next(e for e in errs if e), "failed_ranks": [r for r, e in enumerate(errs) if e]}
    return rec


def main_sharded(args, comm):
    """--mode spatial: ONE solve of --config sharded over the ranks is the headline; step = one CG iteration of that solve (the
    in-library path, csrc/shard.hip: resident CG kernel per rank, device-initiated mailbox stores, no collective in the iteration)."""
    K, W = args.steps, args.warmup
    rec = measure_sharded(args.config, comm, K, W)
    if comm.rank == 0:
        out = {"metric": "cg_matvecs_per_sec", "value": rec["matvecs_per_sec"], "unit": "matvec/s", "n_gpus": comm.world, "steps": K,
               "warmup": W, "ms_per_step": rec["us_per_iteration_device"] * 1e-3, "higher_is_better": True, "scaling": "strong",
               "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": rec["workload"], "parallelism": f"row_slabs{comm.world}"},
               "cg_iters_per_sec": rec["cg_iters_per_sec"], "us_per_iteration_device": rec["us_per_iteration_device"],
               "roofline": {k: rec[k] for k in ("bound", "achieved", "peak", "unit", "frac", "hbm_streaming_equivalent_frac")} |
                           {"kernel": "k_cg_wg<SHARD>", "traffic": None, "ranks": rec["ranks"], "rccl_ranks": rec["rccl_ranks"]},
               "spatial": {args.config: rec}}
        if not args.no_cpu and comm.world >= 1:
            out["cpu_baseline"] = cpu_baseline_leg(args.config, args.cpu_seconds)
        print(json.dumps(out))
    comm.close()


def cpu_baseline_leg(tag, seconds):
    """The CPU oracle (1 thread = the reference's configuration) on a bounded sample of config `tag` with the synthetic model of
    make_sharded: un-preconditioned CG iterations of one right-hand side."""
    try:
        import numpy as np
        from elphdynamics_amd import configs, lattice as lat, synth
        from oracle.oracle import Oracle
        orc = Oracle(fast=True)
        kind, norb, Ls, bonds, beta, dtau = configs.CONFIGS[tag]
        la = lat.Lattice(norb, Ls, Ls if Ls > 1 else 1, 1)
        raw = np.concatenate([la.calc_neighbor_table(o1, o2, d) for (o1, o2, d) in bonds], axis=0)
        cb = lat.initialize_checkerboard(raw, np.ones(raw.shape[0]), dtau)
        N, Ltau = la.nsites, lat.ltau_from_beta(beta, dtau)
        if kind == "holstein":
            om = orc.make_model(0, N, Ltau, cb["table"], cb["cosht"], cb["sinht"], np.exp(-dtau * synth.phonon_field(N, Ltau, beta, dtau)))
        else:
            nb = raw.shape[0]
            tp = 1.0 - 0.1 * 0.25 * synth.phonon_field(nb, Ltau, beta, dtau, omega=0.1, lam=0.0).reshape(nb, Ltau)
            om = orc.make_model(1, N, Ltau, cb["table"], np.cosh(dtau * tp).reshape(-1), np.sinh(dtau * tp).reshape(-1), np.ones(N))
        b0 = synth.rhs(N * Ltau)
        tt = time.perf_counter()
        orc.cg_solve(om, b0, tol=0.0, maxiter=100)
        per_it = (time.perf_counter() - tt) / 100
        n_it = int(max(100, min(200000, seconds / per_it)))
        tt = time.perf_counter()
        orc.cg_solve(om, b0, tol=0.0, maxiter=n_it)
        dt = time.perf_counter() - tt
        return {"value": 2.0 * n_it / dt, "unit": "matvec/s", "cores": 1, "kind": "port",
                "sample": f"{n_it} un-preconditioned CG iterations (tol=0) of one right-hand side of config {tag}, oracle/elph_oracle.c built "
                          f"-O3 -march=native -ffast-math, single thread ({os.cpu_count()} host cores present)",
                "cg_iters_per_sec": n_it / dt, "seconds": dt}
    except Exception as e:
        return {"value": None, "unit": "matvec/s", "cores": 1, "kind": "port", "sample": f"failed: {e}"}


if __name__ == "__main__":
    main()
