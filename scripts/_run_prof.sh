set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "kernels or model or parity" > gpurun_out/r2_t5.log 2>&1; tail -2 gpurun_out/r2_t5.log
rm -rf gpurun_out/prof_r2
mkdir -p gpurun_out/prof_r2
for m in infer train rt x4; do
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r2/$m -- python3 bench.py --steps 10 --warmup 3 --mode $m --no-cpu-baseline > gpurun_out/prof_r2/$m.log 2>&1
  echo "done $m"
done
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_r2/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --mode infer --no-cpu-baseline > gpurun_out/prof_r2/pmc_fetch.log 2>&1
echo "done fetch"
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_r2/pmc_write -- python3 bench.py --steps 3 --warmup 1 --mode infer --no-cpu-baseline > gpurun_out/prof_r2/pmc_write.log 2>&1
echo "done write"
find gpurun_out/prof_r2 -name "*.csv" | head -30
# keep the merge small: drop the per-dispatch traces of the stats runs
find gpurun_out/prof_r2 -name "*kernel_trace.csv" -path "*infer*" -o -name "*kernel_trace.csv" -path "*train*" -o -name "*kernel_trace.csv" -path "*rt/*" -o -name "*kernel_trace.csv" -path "*x4*" | xargs rm -f
du -sh gpurun_out/prof_r2
