#!/bin/bash
# same-box A/B of two builds of the library (libhybrid_hip_old.so / libhybrid_hip_new.so next to libhybrid_hip.so), alternating runs
P=${GRAFT_REPO_ROOT:-/root/repo}/transformer_cnn_hybrid_network_for_video_processing_amd
cd ${GRAFT_REPO_ROOT:-/root/repo}
for v in old new old new; do
  cp $P/libhybrid_hip_$v.so $P/libhybrid_hip.so
  echo -n "$v: "
  python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline --no-pipeline --no-fwd-bwd-only "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), 'clips/s', round(d['ms_per_step'],4), 'ms')"
done
cp $P/libhybrid_hip_new.so $P/libhybrid_hip.so
