#!/bin/bash
# quick kernel-trace summary of the default bench command (graph mode) -> gpurun_out/kt_quick/
set -e
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/kt_quick
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o kt -- python3 $REPO/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-pipeline --no-fwd-bwd-only "$@" > $OUT/bench.json 2> $OUT/err.log
rm -f $OUT/kt_kernel_trace.csv
cut -c1-200 $OUT/bench.json
