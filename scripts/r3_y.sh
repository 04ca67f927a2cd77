#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3y; rm -rf $O; mkdir -p $O
timeout -k 10 300 python3 scripts/host_profile.py ft > $O/host_ft.log 2>&1; head -50 $O/host_ft.log
timeout -k 10 300 python3 scripts/host_profile.py rt > $O/host_rt.log 2>&1; head -12 $O/host_rt.log
echo all done
