set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 200 python scripts/_tail_stamps.py > gpurun_out/r2_tail_stamps.log 2>&1 || true
tail -2 gpurun_out/r2_tail_stamps.log
rm -rf gpurun_out/prof_r2a
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r2a -- python3 bench.py --steps 10 --warmup 3 --mode infer --no-cpu-baseline > gpurun_out/r2_prof_a.log 2>&1
find gpurun_out/prof_r2a -name "*kernel_stats.csv" | head
