# usage: bash gpurun_prof.sh <tag>   (run on the GPU box through gpurun)
set -x
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# (hipGraph replay is opt-in, so the default bench already launches eagerly: bench == profile)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1600 --warmup 160 --no-cpu --no-sweep > $OUT/bench_trace.json 2> $OUT/bench_trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $GRAFT_REPO_ROOT/bench.py --steps 160 --warmup 16 --no-cpu --no-sweep > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $GRAFT_REPO_ROOT/bench.py --steps 160 --warmup 16 --no-cpu --no-sweep > /dev/null 2> $OUT/pmc_write.err
ls -R $OUT | head -40
