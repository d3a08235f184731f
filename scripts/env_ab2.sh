#!/bin/bash
# same-box A/B of environment switches: scripts/env_ab2.sh "A_ENV=.. B_ENV=.." "A_ENV=.. " ...  (each argument: one variant's environment), alternating, 2 rounds
cd ${GRAFT_REPO_ROOT:-/root/repo}
for round in 1 2; do
  for v in "$@"; do
    echo -n "[$v] "
    env $v python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-roofline --no-pipeline --no-fwd-bwd-only --no-extra-legs | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), 'clips/s', round(d['ms_per_step']*1000,1), 'us')"
  done
done
