#!/bin/bash
# Same-box timing of several builds of the library (alternating processes): the streamed kernel's six-block launch at 1,920 windows.
#   bash scripts/ab_libs.sh "" _prev "" _prev      -> transformerupscaler_amd/libtupscale_hip<suffix>.so each
cd $GRAFT_REPO_ROOT
for L in "$@"; do
  echo -n "[$L]: "
  TUP_LIB_PATH=$GRAFT_REPO_ROOT/transformerupscaler_amd/libtupscale_hip$L.so timeout -k 10 120 python3 scripts/ab_stream.py 1920 21 2>&1 | grep "^stream"
done
