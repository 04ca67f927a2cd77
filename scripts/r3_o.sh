#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3o; rm -rf $O; mkdir -p $O
timeout -k 10 300 python3 scripts/ab_attn_bwd.py > $O/ab2.log 2>&1; tail -3 $O/ab2.log
TUP_LIB_PATH=$PWD/transformerupscaler_amd/csrc/build/ab_occ1.so timeout -k 10 300 python3 scripts/ab_attn_bwd.py > $O/ab1.log 2>&1; tail -3 $O/ab1.log
timeout -k 10 600 python3 -m pytest tests/test_hip_kernels.py tests/test_hip_dropout.py tests/test_window_transformer.py -m gpu -q -x > $O/tests.log 2>&1 || (grep -n "^FAILED\|^E " $O/tests.log | head -30; echo TESTS FAILED)
tail -2 $O/tests.log
echo all done
