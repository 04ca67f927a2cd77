set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_train
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_train -- python3 bench.py --steps 40 --warmup 3 --mode train --no-cpu-baseline > gpurun_out/prof_train.log 2>&1
cp $(ls gpurun_out/prof_train/*/*kernel_stats.csv | head -1) gpurun_out/stats_train.csv
