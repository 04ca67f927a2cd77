#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/t; rm -rf $O; mkdir -p $O
B=$PWD/transformerupscaler_amd/csrc/build
timeout -k 10 300 python3 -m pytest tests/test_hip_kernels.py tests/test_hip_train.py -m gpu -q -x -k "planar or train" 2>&1 | tail -2
for v in old new; do
  TUP_LIB_PATH=$B/ab_$v.so timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t_$v -- python3 bench.py --steps 6 --warmup 2 --mode train --no-cpu-baseline > $O/t_$v.log 2>&1
  echo "== $v"; python3 - <<PY
import csv,glob
f=glob.glob('$O/t_$v/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if 'wgrad_planar' in r['Name'] or 'c3_persistent' in r['Name']: print('  %8.1f us x %3d  %s'%(float(r['AverageNs'])/1e3, int(r['Calls']), r['Name'][:70]))
PY
done
rm -rf $O/t_old $O/t_new
