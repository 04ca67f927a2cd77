#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
L=$PWD/transformerupscaler_amd/csrc/build/ab_noat.so
echo base; timeout -k 10 120 python3 scripts/microbench_conv_wgrad.py 2>&1 | grep conv_c64; timeout -k 10 120 python3 scripts/microbench_wgrad.py 2>&1 | grep NI
echo noatomic; TUP_LIB_PATH=$L timeout -k 10 120 python3 scripts/microbench_conv_wgrad.py 2>&1 | grep conv_c64; TUP_LIB_PATH=$L timeout -k 10 120 python3 scripts/microbench_wgrad.py 2>&1 | grep NI
echo all done
