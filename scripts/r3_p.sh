#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3p; rm -rf $O; mkdir -p $O
for s in 64 85 128 170; do
  echo "occ2 slots $s"; TUP_ATTN_BWD_SLOTS=$s timeout -k 10 200 python3 scripts/ab_attn_bwd.py 2>&1 | grep dropout
  echo "occ1 slots $s"; TUP_ATTN_BWD_SLOTS=$s TUP_LIB_PATH=$PWD/transformerupscaler_amd/csrc/build/ab_occ1.so timeout -k 10 200 python3 scripts/ab_attn_bwd.py 2>&1 | grep dropout
done > $O/sweep.log 2>&1
cat $O/sweep.log

timeout -k 10 600 python3 -m pytest tests/test_window_transformer.py -m gpu -q -s > gpurun_out/r3p/wt.log 2>&1 || (grep -n "^FAILED\|^E " gpurun_out/r3p/wt.log | head -30; echo WT TESTS FAILED)
tail -3 gpurun_out/r3p/wt.log
echo all done
