#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3m; rm -rf $O; mkdir -p $O
B=transformerupscaler_amd/csrc/build
timeout -k 10 300 python3 scripts/ab_block.py bufdma=$B/ab_bufdma.so@abi8 new=transformerupscaler_amd/libtupscale_hip.so > $O/ab.log 2>&1; tail -5 $O/ab.log
timeout -k 10 600 python3 -m pytest tests/test_hip_kernels.py -m gpu -q -s -k "block or fused" > $O/tests.log 2>&1 || (grep -n "^FAILED\|^E " $O/tests.log | head -30; echo TESTS FAILED)
tail -2 $O/tests.log
timeout -k 10 200 python3 scripts/_block32_stamps.py 6 > $O/stamps6.log 2>&1; tail -4 $O/stamps6.log
echo all done
