#!/bin/bash
# Same-box timing of the training step (bench.py --mode train, 4 x 720p -> 1080p) with several builds of the library, alternating processes.
#   bash scripts/ab_train_libs.sh "" _prev "" _prev      -> transformerupscaler_amd/libtupscale_hip<suffix>.so each
cd $GRAFT_REPO_ROOT
for L in "$@"; do
  echo -n "[$L]: "
  TUP_LIB_PATH=$GRAFT_REPO_ROOT/transformerupscaler_amd/libtupscale_hip$L.so timeout -k 10 300 python3 bench.py --mode ${MODE:-train} --no-cpu-baseline --no-sustained 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['metric'][:40], round(d['ms_per_step'],3), 'ms', d.get('repetitions_ms_per_step'))"
done
