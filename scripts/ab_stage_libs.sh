#!/bin/bash
# Same-box per-stage timing of the inference forward (scripts/stage_times.py) with several builds of the library, alternating processes.
#   bash scripts/ab_stage_libs.sh "" _nt "" _nt
cd $GRAFT_REPO_ROOT
for L in "$@"; do
  echo "[$L]"
  TUP_LIB_PATH=$GRAFT_REPO_ROOT/transformerupscaler_amd/libtupscale_hip$L.so timeout -k 10 200 python3 scripts/stage_times.py 2>&1 | grep -v amdgpu.ids | tr '\n' ' '; echo
done
