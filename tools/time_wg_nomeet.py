#!/usr/bin/env python3
"""What the meeting of k_cg_wg costs, measured as a bound: the product kernel against a diagnostic build whose iteration has NO meeting
(tools/build_wg_nomeet.sh: no record, no poll, no boundary slices — the most a pipelined recurrence or an XCD-local meeting could save),
at 2 and at 4 slices per wave (ELPH_WG_T), for the batch sizes where the shapes compete.  Config C.
usage: python3 tools/time_wg_nomeet.py          (starts one child per library and shape: ELPH_LIB / ELPH_WG_T are read at load time)"""
import ctypes as C
import json
import os
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
NRS = (24, 48, 96, 288)


def child():
    import numpy as np
    sys.path.insert(0, ROOT)
    from elphdynamics_amd import _lib, configs
    from elphdynamics_amd._lib import check
    lib = _lib.load()
    m = configs.make_model("C", tol=1e-5)
    out = {}
    for nr in NRS:
        _, Bs = configs.rhs(m, nr)
        ms = C.c_double()
        best = 1e30
        for reps in (200, 1000, 1000):
            check(lib.elph_bench_prepare(m._h, 1, nr, _lib.dptr(np.ascontiguousarray(Bs))))
            try:
                check(lib.elph_bench_run(m._h, 9, nr, reps, 0, C.byref(ms)))
            except Exception as e:
                print(f"nrhs {nr} reps {reps} lib {os.environ.get('ELPH_LIB')} T {os.environ.get('ELPH_WG_T')}: {e}", flush=True)
                best = float("nan")
                break
            if reps == 1000:
                best = min(best, 1e3 * ms.value / reps)
        use, T, W, G = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        check(lib.elph_bench_wg_info(m._h, nr, C.byref(use), C.byref(T), C.byref(W), C.byref(G)))
        out[nr] = (best, T.value, W.value, G.value)
    m.close()
    print("RESULT " + json.dumps(out), flush=True)


def main():
    rows = {}
    for T in (2, 4):
        for name, lib in (("product", "libelphgpu.so"), ("no meeting", "libelphgpu_nomeet.so")):
            env = dict(os.environ, ELPH_LIB=os.path.join(ROOT, "elphdynamics_amd", lib), ELPH_WG_T=str(T), ELPH_TIME_WG_CHILD="1")
            p = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True, timeout=600)
            print(p.stdout[-1500:] if "RESULT" not in p.stdout.splitlines()[0:1] else "", end="")
            if p.returncode != 0:
                print(p.stderr[-2000:])
                raise SystemExit(1)
            rows[(T, name)] = json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    print("config C, un-preconditioned CG iteration of the resident kernel, us per iteration of the whole batch (M mat-vecs/s)")
    for nr in NRS:
        for T in (2, 4):
            a, b = rows[(T, "product")][str(nr)], rows[(T, "no meeting")][str(nr)]
            print(f"  nrhs {nr:3d}  T={a[1]} W={a[2]} G={a[3]}:  product {a[0]:7.2f} us ({2*nr/a[0]:5.2f} M)   without the meeting {b[0]:7.2f} us "
                  f"({2*nr/b[0]:5.2f} M)   meeting = {100*(a[0]-b[0])/a[0]:4.1f} % of the iteration")
    a4 = rows[(4, "product")]["288"][0]
    b2 = rows[(2, "no meeting")]["288"][0]
    print(f"  288 right-hand sides: the product (4 slices per wave, with its meeting) {a4:.2f} us; 2 slices per wave with NO meeting and no extra "
          f"recurrences {b2:.2f} us -> a pipelined recurrence at 2 slices per wave {'cannot win' if b2 >= a4 else 'could win at most ' + format(a4 - b2, '.2f') + ' us'}")


if __name__ == "__main__":
    child() if os.environ.get("ELPH_TIME_WG_CHILD") else main()
