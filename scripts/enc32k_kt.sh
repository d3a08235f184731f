#!/bin/bash
# kernel-trace summary of the Encoder_32K bench (2 warm-up + reps training passes, forward passes before them) -> gpurun_out/kt_enc32k/
set -e
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/kt_enc32k
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o kt -- python3 $REPO/scripts/enc32k_bench.py --frames 16 --reps 5 "$@" > $OUT/bench.json 2> $OUT/err.log
rm -f $OUT/kt_kernel_trace.csv
cat $OUT/bench.json
