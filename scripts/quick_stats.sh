#!/bin/bash
# Quick look on the MI355X box (through gpurun): the GPU tests matching $1 (pytest -k), then kernel stats of the bench modes in $2.
#   gpurun -- bash scripts/quick_stats.sh "patch_embed or gemm" "infer train"
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/quick; rm -rf $O; mkdir -p $O
if [ -n "$1" ]; then timeout -k 10 600 python3 -m pytest tests -m gpu -q -x -k "$1" > $O/tests.log 2>&1 || (tail -30 $O/tests.log; echo TESTS FAILED; exit 1); tail -1 $O/tests.log; fi
for m in $2; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$m -- python3 bench.py --steps 10 --warmup 3 --mode $m --no-cpu-baseline > $O/stats_$m.log 2>&1
  cp $(ls $O/stats_$m/*/*kernel_stats.csv | head -1) $O/kernel_stats_$m.csv
  rm -rf $O/stats_$m
  tail -c 400 $O/stats_$m.log; echo
  python3 - $O/kernel_stats_$m.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("total kernel ms", sum(float(r["TotalDurationNs"]) for r in rows) / 1e6)
for r in rows[:28]:
    print("  %5.1f%% %8.1f us x %5d  %s" % (float(r["Percentage"]), float(r["AverageNs"]) / 1e3, int(r["Calls"]), r["Name"][:95]))
PY
done
echo quick done
