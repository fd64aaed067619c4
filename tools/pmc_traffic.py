#!/usr/bin/env python3
"""Per-kernel HBM-side bytes per launch from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of the same command.

usage: python tools/pmc_traffic.py <dir with pmc_fetch/ and pmc_write/>   -> JSON on stdout

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
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
