#!/bin/bash
# A/B of library builds on one MI355X box (through gpurun): kernel stats of bench.py per mode with TUP_LIB_PATH pointing at each build.
#   gpurun -- bash scripts/ab_libs.sh "infer train" "pattern|pattern" name=path.so [name=path.so ...]     (a name may repeat)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/ablibs; rm -rf $O; mkdir -p $O
modes=$1; pat=$2; shift 2
for spec in "$@"; do
  name=${spec%%=*}; lib=${spec#*=}
  for m in $modes; do
    ( export TUP_LIB_PATH=$GRAFT_REPO_ROOT/$lib; timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/run -- python3 bench.py --steps 10 --warmup 3 --mode $m --no-cpu-baseline > $O/$name.$m.log 2>&1 )
    python3 - $(ls $O/run/*/*kernel_stats.csv | head -1) $name $m "$pat" <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print(sys.argv[2], sys.argv[3], 'total %.2f ms |' % (sum(float(r['TotalDurationNs']) for r in rows) / 1e6),
      ' | '.join('%s %.1f' % (r['Name'].split('::')[-1][:30], float(r['AverageNs']) / 1e3) for r in rows if re.search(sys.argv[4], r['Name'])))
PY
    rm -rf $O/run
  done
done
