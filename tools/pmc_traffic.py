#!/usr/bin/env python3
"""Per-kernel HBM-side bytes per launch from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of the same command.

usage: python tools/pmc_traffic.py <dir with pmc_fetch/ and pmc_write/>   -> JSON on stdout
       python tools/pmc_traffic.py <dir> --traffic-json NRHS NDIM > profiles/traffic.json   (what bench.py reads)

rocprofv3 reports both counters in KB per dispatch.  Corrections (MI355X_MICROARCH.md, "HBM"): on gfx950 FETCH_SIZE
tallies 128-byte requests at 64 bytes, so reads are DOUBLED; WRITE_SIZE is exact.  The factor is calibrated on this
code's own access pattern (8 B/lane coalesced f64) with k_cg_xr, whose traffic is known exactly: it reads 4 vectors and
writes 2 (nrhs * Ndim * 8 B each).
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
            k = r["Kernel_Name"]
            a = acc[k]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
            a[2] = int(r.get("Grid_Size", 0) or 0)
    return acc


def main():
    root = sys.argv[1]
    fetch, write = load(os.path.join(root, "pmc_fetch"), "FETCH_SIZE"), load(os.path.join(root, "pmc_write"), "WRITE_SIZE")
    out = []
    for k in sorted(set(fetch) | set(write)):
        f, w = fetch.get(k, [0.0, 0, 0]), write.get(k, [0.0, 0, 0])
        fk = f[0] / max(f[1], 1)
        wk = w[0] / max(w[1], 1)
        rd, wr = 2.0 * fk * 1024.0, wk * 1024.0
        out.append({"kernel": k.split("(")[0], "grid_threads": f[2] or w[2], "dispatches": max(f[1], w[1]), "FETCH_SIZE_KB_raw": fk,
                    "WRITE_SIZE_KB_raw": wk, "hbm_read_bytes_corrected": rd, "hbm_write_bytes": wr, "hbm_bytes_per_launch": rd + wr})
    if len(sys.argv) > 2 and sys.argv[2] == "--traffic-json":
        nrhs, ndim = int(sys.argv[3]), int(sys.argv[4])

        def find(sub):
            return max((r for r in out if sub in r["kernel"]), key=lambda r: r["dispatches"])
        ap, xr = find("k_cg_ap"), find("k_cg_xr")
        vec = ndim * nrhs * 8
        note = ("HBM-side bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/profile_bench.sh + "
                "tools/pmc_traffic.py) on the default bench.py workload (%d right-hand sides; --steps 160). FETCH_SIZE is doubled (gfx950 "
                "counts 128-B requests at 64 B: MI355X_MICROARCH.md HBM section); the factor is calibrated on this access pattern "
                "(8 B/lane coalesced f64): k_cg_xr reads exactly 2 vectors = %.1f MB and FETCH_SIZE*2 reads %.1f MB; WRITE_SIZE is exact "
                "(k_cg_xr writes 1 vector = %.2f MB vs %.2f MB measured)." %
                (nrhs, 2 * vec / 1e6, xr["hbm_read_bytes_corrected"] / 1e6, vec / 1e6, xr["hbm_write_bytes"] / 1e6))
        tj = {"_note": note}
        for key, r in (("k_cg_ap", ap), ("k_cg_xr", xr)):
            tj["%s_nrhs%d" % (key, nrhs)] = {"kernel": r["kernel"], "hbm_bytes_per_launch": r["hbm_bytes_per_launch"],
                                             "read": r["hbm_read_bytes_corrected"], "write": r["hbm_write_bytes"]}
        json.dump(tj, sys.stdout, indent=1)
        return
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
