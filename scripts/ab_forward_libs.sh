#!/bin/bash
# Same-box timing of the inference forward (BASELINE configs[1], per-stage events) with several builds of the library, alternating processes.
#   bash scripts/ab_forward_libs.sh "" _t3 "" _t3
cd $GRAFT_REPO_ROOT
for L in "$@"; do
  echo "[$L]"
  TUP_LIB_PATH=$GRAFT_REPO_ROOT/transformerupscaler_amd/libtupscale_hip$L.so timeout -k 10 200 python3 scripts/stage_times.py 2>&1 | grep "forward\|tail"
done
