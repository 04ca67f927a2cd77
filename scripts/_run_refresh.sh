set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for m in train; do
  rm -rf gpurun_out/prof_$m
  timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$m -- python3 bench.py --steps 10 --warmup 3 --mode $m --no-cpu-baseline > gpurun_out/prof_$m.log 2>&1
  cp $(ls gpurun_out/prof_$m/*/*kernel_stats.csv | head -1) gpurun_out/stats_$m.csv
  echo "$m done"
done
