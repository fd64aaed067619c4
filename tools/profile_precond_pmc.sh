#!/bin/bash
# PMC passes over the KPM-preconditioned batch iteration (bench.py --precond): SQ issue/wait counters, L2 hit/miss, memory-side bytes.
# usage (GPU box): bash tools/profile_precond_pmc.sh <tag> [bench.py flags]     (one rocprofv3 --pmc pass per counter group; no trace domains)
set -uo pipefail
TAG=${1:-precond}
shift || true
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
pass() {  # name, counters...
    local name=$1; shift
    rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $GRAFT_REPO_ROOT/bench.py --precond --steps 32 --warmup 4 --no-cpu --no-sweep --no-spatial "${EXTRA[@]}" > $OUT/$name.json 2> $OUT/$name.err || echo "pass $name failed (rc $?)" >> $OUT/failed.txt
}
EXTRA=("$@")
pass sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY
pass sq2 SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pass fetch FETCH_SIZE
pass write WRITE_SIZE
cd $GRAFT_REPO_ROOT
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        a = acc[k][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
names = sorted({c for k in acc for c in acc[k]})
with open(out + "/counters_per_dispatch.csv", "w") as g:
    g.write("kernel,dispatches," + ",".join(names) + "\n")
    for k in sorted(acc, key=lambda k: -acc[k].get("SQ_WAVE_CYCLES", [0, 1])[0]):
        n = max(v[1] for v in acc[k].values())
        g.write('"%s",%d,' % (k, n) + ",".join(("%.1f" % (acc[k][c][0] / max(acc[k][c][1], 1))) if c in acc[k] else "" for c in names) + "\n")
print(open(out + "/counters_per_dispatch.csv").read()[:6000])
PY
