#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3f; rm -rf $O; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_hip_kernels.py tests/test_hip_model.py tests/test_hip_parity_r2.py tests/test_image_io.py -m gpu -q -x -k "tail or model or config or psnr or golden or fixture or forward or overlay" > $O/tests.log 2>&1; tail -5 $O/tests.log
timeout -k 10 300 python3 bench.py --mode infer --no-cpu-baseline > $O/bench_infer.json 2> $O/bench_infer.err; python3 -c "
import json;d=json.loads(open('$O/bench_infer.json').read().strip().splitlines()[-1]);print('infer',d['value'],d['ms_per_step'])"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_infer -- python3 bench.py --steps 10 --warmup 3 --mode infer --no-cpu-baseline > $O/stats_infer.log 2>&1
cp $(ls $O/stats_infer/*/*kernel_stats.csv | head -1) $O/kernel_stats_infer.csv; rm -rf $O/stats_infer
head -12 $O/kernel_stats_infer.csv | cut -c1-60,150-230
echo all done
