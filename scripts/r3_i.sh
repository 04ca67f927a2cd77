#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3i; rm -rf $O; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_hip_kernels.py tests/test_hip_model.py tests/test_hip_rt.py tests/test_window_transformer.py tests/test_hip_train.py -m gpu -q -x > $O/tests.log 2>&1; tail -3 $O/tests.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_infer -- python3 bench.py --steps 10 --warmup 3 --mode infer --no-cpu-baseline > $O/stats_infer.log 2>&1
cp $(ls $O/stats_infer/*/*kernel_stats.csv | head -1) $O/kernel_stats_infer.csv; rm -rf $O/stats_infer
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r3i/kernel_stats_infer.csv')))
for r in rows[:10]: print(f"{float(r['AverageNs'])/1e3:9.1f} us x{int(r['Calls'])/13:4.1f} {r['Name'][:80]}")
PY
timeout -k 10 300 python3 bench.py --mode infer --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('infer',d['value'],d['ms_per_step'])"
timeout -k 10 300 python3 bench.py --mode train 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('train',d['value'],d['ms_per_step'])"
echo all done
