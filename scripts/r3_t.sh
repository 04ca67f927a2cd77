#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for b in 2048 1024 512 256 128; do TUP_COLSUM_BLOCKS=$b timeout -k 10 120 python3 scripts/microbench_colsum.py 2>&1 | grep target; done
for b in 256 128 64; do TUP_LN_BWD_BLOCKS=$b timeout -k 10 120 python3 scripts/microbench_ln_bwd.py 2>&1 | grep blocks; done
echo all done
