#!/bin/bash
# same-box A/B of environment switches in one precision mode: scripts/env_ab_mode.sh <dtype> "A_ENV=.." "B_ENV=.." ... (alternating, 2 rounds)
cd ${GRAFT_REPO_ROOT:-/root/repo}
M=$1; shift
for round in 1 2; do
  for v in "$@"; do
    echo -n "[$M $v] "
    env $v python bench.py --dtype $M --steps 200 --warmup 20 --no-cpu-baseline --no-roofline --no-pipeline --no-fwd-bwd-only --no-extra-legs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), 'clips/s', round(d['ms_per_step']*1000,1), 'us')"
  done
done
