#!/bin/bash
# A/B of HIP runtime switches that affect hipGraph launch cost: step time of the plain graph-mode bench under each setting
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
run() { echo "== $*"; env "$@" python3 bench.py --steps 60 --warmup 8 --no-extra-legs --no-cpu-baseline --no-roofline --no-pipeline --no-fwd-bwd-only 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('graph_fallback'))"; }
run A=1
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run HIP_FORCE_DEV_KERNARG=0
run HIP_FORCE_DEV_KERNARG=1
run DEBUG_HIP_GRAPH_BATCH_SIZE=1
run DEBUG_HIP_GRAPH_BATCH_SIZE=256
run DEBUG_HIP_FORCE_GRAPH_QUEUES=1
run A=2
