#!/bin/bash
# kernel trace of a short graph-mode bench run -> ordered launch list of one step (scripts/step_trace.py) + per-kernel stats;  args: tag, then bench.py flags
set -e
TAG=${1:-bf16}; shift || true
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/step_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o kt -- python3 $REPO/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline --no-pipeline --no-fwd-bwd-only "$@" > $OUT/bench.json 2> $OUT/err.log
python3 $REPO/scripts/step_trace.py $OUT/kt_kernel_trace.csv > $OUT/step.txt
python3 $REPO/scripts/kt_summary.py $OUT/kt_kernel_stats.csv 15 50 > $OUT/stats.txt
rm -f $OUT/kt_kernel_trace.csv
tail -1 $OUT/step.txt
