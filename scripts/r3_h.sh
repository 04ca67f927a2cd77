#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3h; rm -rf $O; mkdir -p $O
for v in base cheap; do
  if [ $v = cheap ]; then export TUP_LIB_PATH=$GRAFT_REPO_ROOT/transformerupscaler_amd/csrc/build/ab_cheaphash.so; fi
  timeout -k 10 300 python3 bench.py --mode rt > $O/rt_$v.json 2>/dev/null; python3 -c "
import json;d=json.loads(open('$O/rt_$v.json').read().strip().splitlines()[-1]);print('$v rt',d['value'],d['ms_per_step'])"
  timeout -k 10 300 python3 bench.py --mode train > $O/train_$v.json 2>/dev/null; python3 -c "
import json;d=json.loads(open('$O/train_$v.json').read().strip().splitlines()[-1]);print('$v train',d['value'],d['ms_per_step'])"
done
unset TUP_LIB_PATH
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_train -- python3 bench.py --steps 10 --warmup 3 --mode train > $O/stats_train.log 2>&1
cp $(ls $O/stats_train/*/*kernel_stats.csv | head -1) $O/kernel_stats_train.csv; rm -rf $O/stats_train
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_rt -- python3 bench.py --steps 10 --warmup 3 --mode rt > $O/stats_rt.log 2>&1
cp $(ls $O/stats_rt/*/*kernel_stats.csv | head -1) $O/kernel_stats_rt.csv; rm -rf $O/stats_rt
echo all done
