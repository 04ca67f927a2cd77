set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_r2b
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r2b -- python3 bench.py --steps 5 --warmup 2 --mode train > gpurun_out/r2_prof_b.log 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/prof_r2b/*/*kernel_stats.csv")[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total ms/step", tot/1e6/7)
for r in rows[:45]:
    print(f'{float(r["TotalDurationNs"])/1e6/7:7.3f} ms  {int(r["Calls"])/7:5.1f}x {float(r["AverageNs"])/1e3:8.1f} us  {r["Name"][:95]}')
PY
