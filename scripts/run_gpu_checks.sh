#!/bin/bash
# The round's standard GPU check (through gpurun): the GPU test-suite, then the default bench line with its headline numbers.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/checks; rm -rf $O; mkdir -p $O
timeout -k 10 1000 python3 -m pytest tests -m gpu -q -s > $O/tests.log 2>&1 || (grep -n "^FAILED\|^E " $O/tests.log | head -30; echo TESTS FAILED)
tail -2 $O/tests.log
timeout -k 10 400 python3 bench.py > $O/bench.json 2> $O/bench.err
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/checks/bench.json').read().strip().splitlines()[-1])
print('infer', d['value'], d['ms_per_step']); r=d['roofline']; print('roof', r['frac'], r['ms_per_launch'], r['kernel'][:30], r.get('traffic'), (r.get('counters') or {}).get('mfma_util'))
print('train', d['train']['value'], d['train']['ms_per_step']); print('rt', d['rt_train']['value'], d['rt_train']['ms_per_step']); print('x4', d['x4']['value'], d['x4']['ms_per_step'])
PY
echo all done
