#!/bin/bash
# kernel-trace summary of the FCT bench -> gpurun_out/kt_fct/
set -e
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/kt_fct
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o kt -- python3 $REPO/scripts/fct_bench.py --frames 16 --reps 5 "$@" > $OUT/bench.json 2> $OUT/err.log
rm -f $OUT/kt_kernel_trace.csv
cat $OUT/bench.json
