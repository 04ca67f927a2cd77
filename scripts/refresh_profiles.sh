#!/bin/bash
# Regenerates the round's committed evidence on the MI355X box (run through gpurun from the repo root, one part per call: a, then c and b):
#   part a:  profiles/r04_kernel_stats_{infer,train,rt,x4}.csv   rocprofv3 --kernel-trace --stats of bench.py per mode
#            profiles/r04_grid_rounds_{train,rt}.txt             scripts/grid_rounds.py on the same traces
#   part c:  profiles/r04_pmc_mfma{,_train,_rt}.json              two --pmc passes (matrix-pipe / wave-cycle counters) joined by scripts/pmc_mfma.py
#   part b:  profiles/r04_pmc_traffic_{infer,x4,train,rt}.json    two --pmc passes per mode (FETCH_SIZE, WRITE_SIZE) joined by scripts/pmc_traffic.py
#            profiles/r04_block_stream_{ab,stamps}.txt            scripts/ab_stream.py, scripts/bs_stamps.py (diagnostic library)
# Counters are collected in their own runs (--pmc with --kernel-trace only).  Outputs land in gpurun_out/refresh/; copy them to profiles/.
set -e
part=${1:-a}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/refresh
mkdir -p $O
if [ "$part" = a ]; then
  for m in infer train rt x4; do
    rm -rf $O/stats_$m
    timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$m -- python3 bench.py --steps 10 --warmup 3 --mode $m --no-cpu-baseline --no-sustained > $O/stats_$m.log 2>&1
    cp $(ls $O/stats_$m/*/*kernel_stats.csv | head -1) $O/r04_kernel_stats_$m.csv
    python3 scripts/grid_rounds.py $O/stats_$m 200 > $O/r04_grid_rounds_$m.txt 2>&1 || true
    rm -rf $O/stats_$m
    echo "stats $m done"
  done
  timeout -k 10 400 python3 bench.py > $O/r04_bench_default.json 2> $O/bench.err
  tail -c 600 $O/r04_bench_default.json
elif [ "$part" = c ]; then
  # durations: part a's stats (copied to profiles/ between the calls: gpurun_out/ does not travel to the box)
  for m in infer train rt; do cp profiles/r04_kernel_stats_$m.csv $O/r04_kernel_stats_$m.csv; done
  rm -rf $O/pmc_a $O/pmc_b
  timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_a -- python3 bench.py --steps 3 --warmup 1 --mode infer --no-cpu-baseline --no-sustained > $O/pmc_a.log 2>&1
  echo "pmc a done"
  timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_TRANS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_SALU --kernel-trace --output-format csv -d $O/pmc_b -- python3 bench.py --steps 3 --warmup 1 --mode infer --no-cpu-baseline --no-sustained > $O/pmc_b.log 2>&1
  echo "pmc b done"
  python3 scripts/pmc_mfma.py $O/r04_pmc_mfma.json $O/r04_kernel_stats_infer.csv $O/pmc_a $O/pmc_b
  rm -rf $O/pmc_a $O/pmc_b
  # the same two passes for the training steps (VERDICT r3 missing #3: counter evidence outside inference)
  for m in train rt; do
    rm -rf $O/pmc_a $O/pmc_b
    timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_a -- python3 bench.py --steps 3 --warmup 1 --mode $m --no-cpu-baseline --no-sustained > $O/pmc_a_$m.log 2>&1
    timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_TRANS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_SALU --kernel-trace --output-format csv -d $O/pmc_b -- python3 bench.py --steps 3 --warmup 1 --mode $m --no-cpu-baseline --no-sustained > $O/pmc_b_$m.log 2>&1
    python3 scripts/pmc_mfma.py $O/r04_pmc_mfma_$m.json $O/r04_kernel_stats_$m.csv $O/pmc_a $O/pmc_b --mode $m
    echo "pmc mfma $m done"
  done
  rm -rf $O/pmc_a $O/pmc_b
else
  for m in infer x4 train rt; do
    for c in FETCH_SIZE WRITE_SIZE; do
      rm -rf $O/pmc_${m}_$c
      timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_${m}_$c -- python3 bench.py --steps 3 --warmup 1 --mode $m --no-cpu-baseline --no-sustained > $O/pmc_${m}_$c.log 2>&1
      echo "pmc $m $c done"
    done
    # durations: part a's stats of this mode (copied to profiles/ between the two calls: gpurun_out/ does not travel to the box)
    python3 scripts/pmc_traffic.py $O/pmc_${m}_FETCH_SIZE $O/pmc_${m}_WRITE_SIZE profiles/r04_kernel_stats_$m.csv $O/r04_pmc_traffic_$m.json $m
    rm -rf $O/pmc_${m}_FETCH_SIZE $O/pmc_${m}_WRITE_SIZE
  done
fi
if [ "$part" = b ]; then
  timeout -k 10 120 python3 scripts/microbench_hbm.py > $O/r04_microbench_hbm.txt 2>&1; cat $O/r04_microbench_hbm.txt
  # the streamed whole-block kernel: A/B against round 3's kernel in one process, and the per-phase stamps of the diagnostic library
  # (make -C transformerupscaler_amd/csrc diag, before the gpurun call)
  (for n in 1920 960 540 240; do echo "== $n windows"; timeout -k 10 120 python3 scripts/ab_stream.py $n 25 2>&1 | grep -v amdgpu.ids; done) > $O/r04_block_stream_ab.txt
  timeout -k 10 120 python3 scripts/bs_stamps.py 2>&1 | grep -v amdgpu.ids > $O/r04_block_stream_stamps.txt
  tail -4 $O/r04_block_stream_ab.txt
fi
echo "refresh $part done"
