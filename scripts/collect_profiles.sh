#!/bin/bash
# Round profile collection on the GPU box (run through gpurun): kernel-trace summaries of the bench command for configs 2/4/5, then the
# two PMC passes (FETCH_SIZE, WRITE_SIZE; counters only, no trace domains) of config 2.  Output under gpurun_out/prof_<tag>/.
set -e
TAG=${1:-r02}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in 2 4 5; do
  STEPS=30; [ $C != 2 ] && STEPS=12
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_c$C -o kt -- python3 $REPO/bench.py --config $C --steps $STEPS --warmup 5 --no-cpu-baseline > $OUT/bench_c${C}_under_rocprof.json 2> $OUT/kt_c$C.err
  echo "kernel trace config $C done"
done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- python3 $REPO/bench.py --eager --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-fwd-bwd-only --no-pipeline > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err
echo "pmc fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- python3 $REPO/bench.py --eager --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-fwd-bwd-only --no-pipeline > $OUT/pmc_write.json 2> $OUT/pmc_write.err
echo "pmc write done"
find $OUT -name "*.csv" -size +20M -delete     # raw per-dispatch traces are too large to bring back; the stats summaries stay
ls -la $OUT $OUT/*
