#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/final; rm -rf $O; mkdir -p $O
timeout -k 10 1000 python3 -m pytest tests -m gpu -q > $O/tests.log 2>&1 || (grep -n "^FAILED\|^E " $O/tests.log | head -30; echo TESTS FAILED)
tail -2 $O/tests.log
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout -k 10 400 python3 bench.py > $O/bench.json 2> $O/bench.err
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/final/bench.json').read().strip().splitlines()[-1])
print('infer', d['value'], d['ms_per_step']); r=d['roofline']; print('roof', r['frac'], r['ms_per_launch'], r.get('traffic'), (r.get('counters') or {}).get('mfma_util'))
print('train', d['train']['value'], d['train']['ms_per_step']); print('rt', d['rt_train']['value'], d['rt_train']['ms_per_step']); print('x4', d['x4']['value'], d['x4']['ms_per_step'], d['x4']['diagnostics'])
print('overlay', d['overlay']['graph']['p50'])
PY
echo all done
