set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -m pytest tests/test_hip_kernels.py tests/test_hip_model.py -m gpu -q -p no:cacheprovider -k "tail or fixture or full_size" 2>&1 | tail -2
for v in "" "TUP_TAIL_OCC3=1" "TUP_TAIL_OCC2=1"; do
rm -rf gpurun_out/prof_r2a
env $v timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r2a -- python3 bench.py --steps 10 --warmup 3 --mode infer --no-cpu-baseline > gpurun_out/r2_prof_a.log 2>&1
echo "variant: $v"
python3 - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/prof_r2a/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:9]:
    if "tail" in r["Name"]: print(f'{float(r["AverageNs"])/1e3:8.1f} us x {int(r["Calls"])/13:4.1f}  {r["Name"][:90]}')
PY
done
