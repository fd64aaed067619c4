#!/bin/bash
# usage (on the GPU box, through gpurun):  bash tools/profile_bench.sh <tag> [extra bench.py flags, e.g. --nrhs 128 --chains 64]
# SEPARATE rocprofv3 runs of the same bench.py command (kernel trace; PMC FETCH_SIZE; PMC WRITE_SIZE — the two TCC counters
# do not fit one pass and gpurun refuses --pmc combined with trace domains), then the preconditioned batch.
# (--warmup 0: the headline is ONE launch of the workgroup-resident kernel per run; a warm-up launch of another length would only blur its
# average in the kernel statistics.)
# bench.py is ALWAYS started as `python3 bench.py` directly after `--`: never through its shebang, env, taskset or bash -c
# (the profiler's preloaded library has initialised the GPU by then; an exec hop would take the box down).
set -euo pipefail
set -x
TAG=${1:-r02}
shift || true
EXTRA="$*"
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $B --steps 1600 --warmup 0 --no-cpu --no-sweep $EXTRA > $OUT/bench_trace.json 2> $OUT/bench_trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $B --steps 160 --warmup 0 --no-cpu --no-sweep $EXTRA > $OUT/bench_pmc.json 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $B --steps 160 --warmup 0 --no-cpu --no-sweep $EXTRA > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_kpm -- python3 $B --precond --steps 320 --warmup 32 --no-cpu --no-sweep $EXTRA > $OUT/bench_trace_kpm.json 2> $OUT/bench_trace_kpm.err
for d in trace pmc_fetch pmc_write trace_kpm; do
    n=$(find $OUT/$d -name '*.csv' | wc -l)
    if [ "$n" -lt 1 ]; then echo "profile_bench: rocprofv3 pass '$d' produced no csv (see $OUT/*.err)" >&2; exit 1; fi
done
cd $GRAFT_REPO_ROOT
python3 tools/pmc_traffic.py $OUT > $OUT/pmc_traffic.json
cp profiles/traffic.json $OUT/traffic.json 2>/dev/null || true
python3 tools/pmc_traffic.py $OUT --merge $OUT/traffic.json
ls -R $OUT | head -40
