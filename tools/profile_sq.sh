#!/bin/bash
# SQ counters (issue / wait / LDS) of whatever kernels one bench.py command launches — evidence for where a kernel's time goes.
# usage (GPU box): bash tools/profile_sq.sh <tag> [bench.py flags, e.g. --steps 400 --nrhs 24 --chains 12 | --precond --steps 64]
# One rocprofv3 --pmc pass (no trace domains: gpurun refuses the combination); bench.py directly after `--`.
set -euo pipefail
TAG=${1:-sq}
shift || true
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_sq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_WAIT_ANY --output-format csv -d $OUT/pmc_sq -- python3 $GRAFT_REPO_ROOT/bench.py --warmup 0 --no-cpu --no-sweep "$@" > $OUT/bench.json 2> $OUT/pmc.err
cd $GRAFT_REPO_ROOT
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(out + "/pmc_sq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        a = acc[k][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
with open(out + "/sq_counters_per_dispatch.csv", "w") as g:
    names = sorted({c for k in acc for c in acc[k]})
    g.write("kernel,dispatches," + ",".join(names) + "\n")
    for k in sorted(acc, key=lambda k: -acc[k].get("SQ_WAVE_CYCLES", [0, 1])[0]):
        n = max(v[1] for v in acc[k].values())
        g.write('"%s",%d,' % (k, n) + ",".join("%.1f" % (acc[k][c][0] / max(acc[k][c][1], 1)) for c in names) + "\n")
print(open(out + "/sq_counters_per_dispatch.csv").read()[:3000])
PY
