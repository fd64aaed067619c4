# usage (on the GPU box, through gpurun):  bash tools/profile_bench.sh <tag>
# Three SEPARATE rocprofv3 runs of the same bench.py command (kernel trace; PMC FETCH_SIZE; PMC WRITE_SIZE — the two TCC
# counters do not fit one pass and gpurun refuses --pmc combined with trace domains), then the preconditioned batch.
set -x
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $B --steps 1600 --warmup 160 --no-cpu --no-sweep > $OUT/bench_trace.json 2> $OUT/bench_trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $B --steps 160 --warmup 16 --no-cpu --no-sweep > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $B --steps 160 --warmup 16 --no-cpu --no-sweep > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_kpm -- python3 $B --precond --steps 320 --warmup 32 --no-cpu --no-sweep > $OUT/bench_trace_kpm.json 2> $OUT/bench_trace_kpm.err
cd $GRAFT_REPO_ROOT
python3 tools/pmc_traffic.py $OUT > $OUT/pmc_traffic.json
python3 tools/pmc_traffic.py $OUT --traffic-json 128 40960 > $OUT/traffic.json
ls -R $OUT | head -40
