#!/bin/bash
# same box: one graph per step vs three (HYB_GRAPH_SINGLE), twice each
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
run() { echo "== $*"; env "$@" python3 bench.py --steps 200 --warmup 10 --no-extra-legs --no-cpu-baseline --no-roofline --no-pipeline --no-fwd-bwd-only 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('graph_fallback'))"; }
run HYB_GRAPH_SINGLE=1
run HYB_GRAPH_SINGLE=0
run HYB_GRAPH_SINGLE=1
run HYB_GRAPH_SINGLE=0
run HYB_GRAPH_SINGLE=1 DEBUG_HIP_GRAPH_BATCH_SIZE=1024
run HYB_GRAPH_SINGLE=1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
