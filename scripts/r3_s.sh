#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for b in 1280 960 768 640 480 384 256; do TUP_LN_BWD_BLOCKS=$b timeout -k 10 120 python3 scripts/microbench_ln_bwd.py 2>&1 | grep blocks; done
echo all done
