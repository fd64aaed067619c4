"""Re-times, on the box it runs on, the neighbours of every decision the library takes from fitted constants, and prints the
disagreements: (1) un-preconditioned solves — the resident kernel's slices per wave and resident-vs-streaming (cg_wg.hip: pick_shape,
wg_cost_table), per BASELINE config and batch size around the crossovers; (2) the KPM-preconditioned batch — slices per wave of the
p/x-fused k_cg_ap (cg_fast_impl.inc: elph_choose_T_px) and one stream vs two half-batches (elph_api.hip: split_wanted).
A decision is flagged when an alternative is more than 8 % faster than what the library picks by itself.

    python tools/validate_form_choice.py [--quick] > profiles/rNN/validate_form_choice.log        (GPU box)

Every alternative runs in a CHILD process (the pins ELPH_WG_T / ELPH_CHUNK_T are read when a handle is made, ELPH_NO_WG / ELPH_WG_ALWAYS per
solve): one process, one configuration."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import ctypes as C, json, os, sys
import numpy as np
sys.path.insert(0, %r)
from elphdynamics_amd import _lib, configs, models, preconditioners as pc, synth
tag, nrhs, what = sys.argv[1], int(sys.argv[2]), sys.argv[3]
lib = _lib.load()
m = configs.make_model(tag, tol=1e-5)
nch = max(1, nrhs // 2)
if nrhs >= 2:
    if m.kind == 0:
        X = np.stack([synth.phonon_field(m.Nph, m.Ltau, m.beta, m.dtau, seed=100 + c) for c in range(nch)])
    else:
        X = np.stack([m.x * (0.6 + 0.8 * c / nch) * (1.0 + 0.2 * synth.randn(100 + c, m.Ndof)) for c in range(nch)])
    models.update_model_chains_(m, X)
_, B = configs.rhs(m, nrhs)
def run(w, reps):
    ms = C.c_double()
    _lib.check(lib.elph_bench_run(m._h, w, nrhs, reps, 0, C.byref(ms)))
    return 1e3 * ms.value / reps
out = {}
if what == "plain":
    _lib.check(lib.elph_bench_prepare(m._h, 1, nrhs, _lib.dptr(np.ascontiguousarray(B))))
    us, T, W, G = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    _lib.check(lib.elph_bench_wg_info(m._h, nrhs, C.byref(us), C.byref(T), C.byref(W), C.byref(G)))
    out["resident_usable"] = us.value; out["T"] = T.value
    if os.environ.get("ELPH_NO_WG") == "1" or not us.value:
        run(1, 20); out["us"] = run(1, 200); out["form"] = "streaming"
    else:
        run(9, 40); _lib.check(lib.elph_bench_prepare(m._h, 1, nrhs, None)); out["us"] = run(9, 400); out["form"] = "resident T=%%d" %% T.value
else:
    P = pc.SymmetricKPMPreconditioner(m, 20, 0.05, 1.0, 1.0)
    (pc.setup_chains_ if nrhs >= 2 else pc.setup_)(P, rng=np.random.default_rng(7))
    _lib.check(lib.elph_bench_prepare(m._h, 3, nrhs, _lib.dptr(np.ascontiguousarray(B))))
    T = C.c_int(); _lib.check(lib.elph_bench_info(m._h, nrhs, C.byref(T))); out["T"] = T.value
    f = C.c_int(); _lib.check(lib.elph_bench_px_info(m._h, C.byref(f))); out["px"] = f.value
    w = 11 if what == "prec2" else 3
    try:
        run(w, 16); _lib.check(lib.elph_bench_prepare(m._h, 3, nrhs, None)); out["us"] = run(w, 160)
    except Exception as e:
        out["us"] = None; out["err"] = repr(e)[:80]
print("RESULT " + json.dumps(out))
''' % ROOT


def child(tag, nrhs, what, env):
    e = dict(os.environ)
    for k in ("ELPH_WG_T", "ELPH_NO_WG", "ELPH_WG_ALWAYS", "ELPH_CHUNK_T", "ELPH_SPLIT_STREAMS", "ELPH_FUSE_PX"):
        e.pop(k, None)
    e.update(env)
    p = subprocess.run([sys.executable, "-c", CHILD, tag, str(nrhs), what], env=e, capture_output=True, text=True, timeout=600)
    for line in p.stdout.splitlines():
        if line.startswith("RESULT "):
            return json.loads(line[7:])
    return {"us": None, "err": (p.stderr or p.stdout)[-200:]}


def main():
    quick = "--quick" in sys.argv
    flagged = []
    print("== un-preconditioned solves: resident kernel (slices per wave) vs streaming, us per CG iteration of the batch")
    for tag, sizes in (("C", (1, 8, 9, 24, 25, 48, 49, 96, 288)), ("D", (1, 16, 17, 24, 48, 96, 288)), ("E", (1, 8, 9, 64, 256)), ("B", (1, 64, 256))):
        for n in (sizes[::2] if quick else sizes):
            auto = child(tag, n, "plain", {})
            alts = {"streaming": child(tag, n, "plain", {"ELPH_NO_WG": "1"})}
            for T in (1, 2, 4):
                r = child(tag, n, "plain", {"ELPH_WG_T": str(T), "ELPH_WG_ALWAYS": "1"})
                if r.get("us") and r.get("T") == T:
                    alts[f"resident T={T}"] = r
            best = min(((k, v["us"]) for k, v in alts.items() if v.get("us")), key=lambda kv: kv[1])
            mark = ""
            if auto.get("us") and best[1] < 0.92 * auto["us"]:
                mark = f"   <-- {best[0]} is {100 * (1 - best[1] / auto['us']):.0f} % faster"
                flagged.append((tag, n, auto.get("form"), best))
            print(f"{tag} nrhs {n:4d}: library picks {auto.get('form')!s:16s} {auto.get('us') or float('nan'):8.2f} | " +
                  ", ".join(f"{k} {v['us']:.2f}" for k, v in alts.items() if v.get("us")) + mark, flush=True)
    print("== KPM-preconditioned batch: slices per wave of k_cg_ap (p/x-fused) and one stream vs two, us per iteration")
    for tag, sizes in (("C", (32, 64, 128, 192, 256, 288)), ("D", (64, 256)), ("E", (64, 256))):
        for n in (sizes[::2] if quick else sizes):
            auto1 = child(tag, n, "prec", {"ELPH_SPLIT_STREAMS": "0"})
            auto2 = child(tag, n, "prec2", {})
            lib_pick = auto2 if (n >= (64 if tag == "D" else 192) and auto2.get("us")) else auto1      # (elph_api.hip: split_wanted)
            alts = {}
            for T in (8, 16, 20):
                for w in ("prec", "prec2"):
                    r = child(tag, n, w, {"ELPH_CHUNK_T": str(T), "ELPH_SPLIT_STREAMS": "0" if w == "prec" else "1"})
                    if r.get("us") and r.get("T") == T:
                        alts[f"T={T} {'two streams' if w == 'prec2' else 'one stream'}"] = r
            best = min(((k, v["us"]) for k, v in alts.items()), key=lambda kv: kv[1]) if alts else ("-", float("nan"))
            mark = ""
            if lib_pick.get("us") and best[1] < 0.92 * lib_pick["us"]:
                mark = f"   <-- {best[0]} is {100 * (1 - best[1] / lib_pick['us']):.0f} % faster"
                flagged.append((tag, n, "preconditioned", best))
            print(f"{tag} nrhs {n:4d}: library T={lib_pick.get('T')} px={lib_pick.get('px')} {'two streams' if lib_pick is auto2 else 'one stream'} "
                  f"{lib_pick.get('us') or float('nan'):8.2f} (one stream {auto1.get('us') or float('nan'):.2f}, two {auto2.get('us') or float('nan'):.2f}) | " +
                  ", ".join(f"{k} {v['us']:.2f}" for k, v in sorted(alts.items())) + mark, flush=True)
    print("== lattices beyond 320 sites, one right-hand side from x = 0: the slab form (slabs.hip) vs the streaming pair, us per CG iteration (Ltau = 160)")
    sweep = os.path.join(ROOT, "tools", "sweep_slabs.py")
    for Ls in ((24, 32) if quick else (18, 20, 24, 28, 30, 32)):
        res = {}
        for mode, env in (("library", {}), ("streaming", {"ELPH_SLABS": "0"}), ("slabs forced", {"ELPH_SLABS": "1"})):
            e = dict(os.environ, ELPH_WG_TIMEOUT_MS="500")
            for k in ("ELPH_SLABS", "ELPH_SLABS_P"):
                e.pop(k, None)
            e.update(env)
            p = subprocess.run([sys.executable, sweep, str(Ls), "0"], env=e, capture_output=True, text=True, timeout=300)
            for line in p.stdout.splitlines():
                if line.startswith("RESULT"):
                    d = eval(line.split(None, 3)[3])
                    # what = 12 is the slab form where the mode admits it; the library's own pick: the slab form if its rule takes the lattice
                    res[mode] = d
        lib_us = res.get("library", {}).get(12)
        lib_form = "slabs" if isinstance(lib_us, float) else "streaming"
        if lib_form == "streaming":
            lib_us = res.get("library", {}).get(1)
        alts = {"streaming": res.get("streaming", {}).get(1), "slabs": res.get("slabs forced", {}).get(12)}
        alts = {k: v for k, v in alts.items() if isinstance(v, float)}
        best = min(alts.items(), key=lambda kv: kv[1])
        mark = ""
        if isinstance(lib_us, float) and best[1] < 0.92 * lib_us:
            mark = f"   <-- {best[0]} is {100 * (1 - best[1] / lib_us):.0f} % faster"
            flagged.append((f"square {Ls}x{Ls}", 1, lib_form, best))
        print(f"square {Ls} x {Ls}: library picks {lib_form:9s} {lib_us if isinstance(lib_us, float) else float('nan'):7.2f} | " + ", ".join(f"{k} {v:.2f}" for k, v in alts.items()) + mark, flush=True)
    print(f"== {len(flagged)} decision(s) flagged (> 8 % slower than an alternative on this box)")
    for f in flagged:
        print("   ", f)


if __name__ == "__main__":
    main()
