#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3v; rm -rf $O; mkdir -p $O
timeout -k 10 200 python3 scripts/ab_attn_bwd.py 2>&1 | grep dropout
timeout -k 10 900 python3 -m pytest tests -m gpu -q -s > $O/tests.log 2>&1 || (grep -n "^FAILED\|^E " $O/tests.log | head -30; echo TESTS FAILED)
tail -2 $O/tests.log
timeout -k 10 400 python3 bench.py > $O/bench.json 2> $O/bench.err
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r3v/bench.json').read().strip().splitlines()[-1])
print('infer', d['value'], d['ms_per_step']); r=d['roofline']; print('roof', r['frac'], r['ms_per_launch'], r['kernel'][:30])
print('train', d['train']['value'], d['train']['ms_per_step']); print('rt', d['rt_train']['value'], d['rt_train']['ms_per_step']); print('x4', d['x4']['value'], d['x4']['ms_per_step'])
PY
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_train -- python3 bench.py --steps 10 --warmup 3 --mode train --no-cpu-baseline > $O/trace_train.log 2>&1
cp $(ls $O/trace_train/*/*kernel_stats.csv | head -1) $O/kernel_stats_train.csv
python3 scripts/grid_rounds.py $O/trace_train 300 > $O/grid_rounds_train.txt 2>&1; head -50 $O/grid_rounds_train.txt
rm -rf $O/trace_train
echo all done
