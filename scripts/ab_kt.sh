#!/bin/bash
# kernel-trace of the bench under two library builds on the same box -> gpurun_out/kt_ab/{old,new}
P=${GRAFT_REPO_ROOT:-/root/repo}/transformer_cnn_hybrid_network_for_video_processing_amd
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for v in old new; do
  cp $P/libhybrid_hip_$v.so $P/libhybrid_hip.so
  OUT=$REPO/gpurun_out/kt_ab/$v; rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o kt -- python3 $REPO/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-pipeline --no-fwd-bwd-only > $OUT/bench.json 2> $OUT/err.log
  rm -f $OUT/kt_kernel_trace.csv
done
cp $P/libhybrid_hip_new.so $P/libhybrid_hip.so
