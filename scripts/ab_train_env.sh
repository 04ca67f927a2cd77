#!/bin/bash
# Same-box timing of the training step with an environment switch on / off, alternating processes:  bash scripts/ab_train_env.sh TUP_EXP_SIDE 0 1 0 1
cd $GRAFT_REPO_ROOT
v=$1; shift
for x in "$@"; do
  echo -n "[$v=$x]: "
  env $v=$x timeout -k 10 300 python3 bench.py --mode ${MODE:-train} --no-cpu-baseline --no-sustained 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['metric'][:40], round(d['ms_per_step'],3), 'ms', d.get('repetitions_ms_per_step'))"
done
