#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
B=$PWD/transformerupscaler_amd/csrc/build
for v in cur dkvexpf cur dkvexpf; do TUP_LIB_PATH=$B/ab_$v.so timeout -k 10 120 python3 scripts/ab_rt_attn_bwd.py 2>&1 | grep bwd; done
echo all done
