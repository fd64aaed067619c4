#!/usr/bin/env python3
"""Per-kernel HBM-side bytes per launch from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of the same bench.py command.

usage: python tools/pmc_traffic.py <dir>                      -> per-kernel JSON on stdout
       python tools/pmc_traffic.py <dir> --merge profiles/traffic.json
           <dir> holds pmc_fetch/, pmc_write/ (rocprofv3 output) and bench_pmc.json (the bench line printed by the FETCH pass:
           its roofline.traffic_key names kernel, config, nrhs, chains).  The entry for that key is added to / replaced in
           profiles/traffic.json, which bench.py reads for roofline.traffic — and only for an exactly matching key.

rocprofv3 reports both counters in KB per dispatch.  Corrections (MI355X_MICROARCH.md, "HBM"): on gfx950 FETCH_SIZE
tallies 128-byte requests at 64 bytes, so reads are DOUBLED; WRITE_SIZE is exact.  The factor is calibrated on this
code's own access pattern (8 B/lane coalesced f64) with k_cg_xr, whose traffic is known exactly: it reads 2 vectors and
writes 1 (nrhs * Ndim * 8 B each); the calibration is written into the entry.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def load(d, counter):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    acc = defaultdict(lambda: [0.0, 0, 0])
    for f in files:
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            a = acc[r["Kernel_Name"]]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
            a[2] = int(r.get("Grid_Size", 0) or 0)
    return acc


def per_kernel(root):
    fetch, write = load(os.path.join(root, "pmc_fetch"), "FETCH_SIZE"), load(os.path.join(root, "pmc_write"), "WRITE_SIZE")
    if not fetch or not write:
        sys.exit(f"pmc_traffic: no FETCH_SIZE / WRITE_SIZE rows under {root}/pmc_fetch, {root}/pmc_write — a rocprofv3 pass failed")
    out = []
    for k in sorted(set(fetch) | set(write)):
        f, w = fetch.get(k, [0.0, 0, 0]), write.get(k, [0.0, 0, 0])
        fk, wk = f[0] / max(f[1], 1), w[0] / max(w[1], 1)
        rd, wr = 2.0 * fk * 1024.0, wk * 1024.0
        out.append({"kernel": k.split("(")[0], "grid_threads": f[2] or w[2], "dispatches": max(f[1], w[1]), "FETCH_SIZE_KB_raw": fk,
                    "WRITE_SIZE_KB_raw": wk, "hbm_read_bytes_corrected": rd, "hbm_write_bytes": wr, "hbm_bytes_per_launch": rd + wr})
    return out


def main():
    root = sys.argv[1]
    out = per_kernel(root)
    if len(sys.argv) > 3 and sys.argv[2] == "--merge":
        dst = sys.argv[3]
        line = [ln for ln in open(os.path.join(root, "bench_pmc.json")) if ln.startswith("{")][-1]
        bench = json.loads(line)
        rl, cfg = bench["roofline"], bench["config"]
        wg = None
        if "roofline_streaming" in bench:          # headline = workgroup-resident kernel: record it too, then the streaming pair
            wg, rl = rl, bench["roofline_streaming"]
        key, nrhs, ndim = rl["traffic_key"], cfg["nrhs"], cfg["ndim"]
        T = key.split("<T=")[1].split(">")[0] if "<T=" in key else None

        def find(sub, must=None):
            c = [r for r in out if sub in r["kernel"] and (must is None or must in r["kernel"])]
            if not c:
                sys.exit(f"pmc_traffic: no kernel matching '{sub}' / '{must}' in the PMC passes")
            return max(c, key=lambda r: r["dispatches"])
        ap = find("k_cg_ap", f", {T}," if T else "k_cg_ap_fast")
        xr = find("k_cg_xr")
        vec = ndim * nrhs * 8
        tj = {"_note": "HBM-side bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes of the same bench.py "
                       "command: tools/profile_bench.sh + tools/pmc_traffic.py).  FETCH_SIZE is doubled (gfx950 counts 128-B requests "
                       "at 64 B: MI355X_MICROARCH.md, HBM section); each entry carries its own calibration on k_cg_xr, which reads "
                       "exactly 2 vectors and writes 1.  bench.py prints roofline.traffic only for an exactly matching key.",
              "kernels": {}}
        if os.path.exists(dst):
            try:
                old = json.load(open(dst))
                tj["kernels"].update(old.get("kernels", {}))
            except Exception:
                pass
        tj["kernels"][key] = {
            "rocprof_kernel": ap["kernel"], "hbm_bytes_per_launch": ap["hbm_bytes_per_launch"], "read": ap["hbm_read_bytes_corrected"],
            "write": ap["hbm_write_bytes"], "dispatches": ap["dispatches"], "nrhs": nrhs, "ndim": ndim,
            "working_set_MB_per_iteration": (5 * vec + (rl["bytes_per_launch"] - 6 * vec)) / 1e6,   # r, p x2, z, x + the tables
            "calibration_k_cg_xr": {"read_expected": 2 * vec, "read_measured": xr["hbm_read_bytes_corrected"],
                                    "write_expected": vec, "write_measured": xr["hbm_write_bytes"]},
            "k_cg_xr_hbm_bytes_per_launch": xr["hbm_bytes_per_launch"],
        }
        if wg is not None:
            kw = find("k_cg_wg")
            tj["kernels"][wg["traffic_key"].split("|iters=")[0]] = {
                "rocprof_kernel": kw["kernel"], "hbm_bytes_per_launch": kw["hbm_bytes_per_launch"], "read": kw["hbm_read_bytes_corrected"],
                "write": kw["hbm_write_bytes"], "dispatches": kw["dispatches"], "nrhs": nrhs, "ndim": ndim,
                "iterations_per_launch": wg["iterations_per_launch"],
                "hbm_bytes_per_iteration": kw["hbm_bytes_per_launch"] / wg["iterations_per_launch"],
                "note": "one launch = iterations_per_launch CG iterations of all right-hand sides; the Krylov vectors stay on chip: what the "
                        "memory-side counters see are the write-through record / boundary granules of the team meetings and the polls of them",
            }
            print(json.dumps(tj["kernels"][wg["traffic_key"].split("|iters=")[0]], indent=1))
        json.dump(tj, open(dst, "w"), indent=1)
        print(json.dumps(tj["kernels"][key], indent=1))
        return
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
