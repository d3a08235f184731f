#!/bin/bash
# bf16x3 step time under A/B environment switches (same box)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
run() { echo "== $*"; env "$@" python3 bench.py --dtype bf16x3 --steps 30 --warmup 3 --no-extra-legs --no-cpu-baseline --no-roofline --no-pipeline --no-fwd-bwd-only 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
run A=1
run HYB_WGRAD1_WGS=256
run HYB_WGRAD1_WGS=384
run HYB_WGRAD1_WGS=768
run A=2
