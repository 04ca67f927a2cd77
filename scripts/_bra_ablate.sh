#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/bra; rm -rf $O; mkdir -p $O
for a in 0 8 3 11 0; do
  TUP_BRA_ABLATE=$a timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s$a -- python3 bench.py --steps 10 --warmup 3 --mode infer --no-cpu-baseline > $O/s$a.log 2>&1
  python3 - $(ls $O/s$a/*/*kernel_stats.csv | head -1) $a <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'bra_rows' in r['Name'] or 'conv5x5' in r['Name']: print('ablate', sys.argv[2], '%.1f us' % (float(r['AverageNs']) / 1e3), r['Name'][:50])
PY
  rm -rf $O/s$a
done
