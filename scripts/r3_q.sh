#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3q; rm -rf $O; mkdir -p $O
(
for s in 84 128 172; do
  echo "occ2-lds slots $s"; TUP_ATTN_BWD_SLOTS=$s timeout -k 10 200 python3 scripts/ab_attn_bwd.py 2>&1 | grep dropout
done
for s in 85; do
  echo "occ1b slots $s"; TUP_ATTN_BWD_SLOTS=$s TUP_LIB_PATH=$PWD/transformerupscaler_amd/csrc/build/ab_occ1b.so timeout -k 10 200 python3 scripts/ab_attn_bwd.py 2>&1 | grep dropout
done ) > $O/sweep.log 2>&1
cat $O/sweep.log
timeout -k 10 600 python3 -m pytest tests/test_hip_kernels.py tests/test_hip_dropout.py tests/test_window_transformer.py -m gpu -q -x > $O/tests.log 2>&1 || (grep -n "^FAILED\|^E " $O/tests.log | head -30; echo TESTS FAILED)
tail -2 $O/tests.log
echo all done
