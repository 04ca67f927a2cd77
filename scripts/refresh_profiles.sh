#!/bin/bash
# Regenerates the round's committed evidence on the MI355X box (run through gpurun from the repo root):
#   profiles/r02_kernel_stats_{infer,train,rt,x4}.csv   rocprofv3 --kernel-trace --stats of bench.py per mode
#   profiles/r02_pmc_traffic.json                        two --pmc passes (FETCH_SIZE, WRITE_SIZE) joined by scripts/pmc_traffic.py
# Counters are collected in their own runs (--pmc with --kernel-trace only).  Outputs land in gpurun_out/refresh/.
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/refresh
rm -rf $O && mkdir -p $O
for m in infer train rt x4; do
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$m -- python3 bench.py --steps 10 --warmup 3 --mode $m --no-cpu-baseline > $O/stats_$m.log 2>&1
  cp $(ls $O/stats_$m/*/*kernel_stats.csv | head -1) $O/r02_kernel_stats_$m.csv
  echo "stats $m done"
done
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- python3 bench.py --steps 3 --warmup 1 --mode infer --no-cpu-baseline > $O/pmc_$c.log 2>&1
  echo "pmc $c done"
done
python3 scripts/pmc_traffic.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/r02_kernel_stats_infer.csv $O/r02_pmc_traffic.json
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/stats_infer $O/stats_train $O/stats_rt $O/stats_x4
echo refresh done
