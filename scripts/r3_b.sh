#!/bin/bash
set +e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3b
rm -rf $O && mkdir -p $O
B=transformerupscaler_amd/csrc/build
timeout -k 10 300 python3 scripts/ab_block.py base=$B/ab_base.so g16=$B/ab_g16.so@gelu16 > $O/ab_g16.log 2>&1; tail -12 $O/ab_g16.log
timeout -k 10 300 python3 scripts/ab_block_model.py > $O/ab_model.log 2>&1; tail -8 $O/ab_model.log
echo all done
