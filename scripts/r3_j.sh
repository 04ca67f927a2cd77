#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3j; rm -rf $O; mkdir -p $O
B=transformerupscaler_amd/csrc/build
timeout -k 10 300 python3 scripts/ab_block.py base=$B/ab_base.so bufdma=$B/ab_bufdma.so > $O/ab.log 2>&1; tail -9 $O/ab.log
