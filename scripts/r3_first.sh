#!/bin/bash
# round-3 first GPU pass: GPU test-suite, default bench line, MFMA counter passes of the inference mode
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3a
rm -rf $O && mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -m gpu -q -s > $O/tests.log 2>&1 || (grep -n "^FAILED\|^E " $O/tests.log | head -40; echo TESTS FAILED)
tail -3 $O/tests.log
timeout -k 10 400 python3 bench.py > $O/bench.json 2> $O/bench.err
echo bench done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_infer -- python3 bench.py --steps 10 --warmup 3 --mode infer --no-cpu-baseline > $O/stats_infer.log 2>&1
cp $(ls $O/stats_infer/*/*kernel_stats.csv | head -1) $O/kernel_stats_infer.csv
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_a -- python3 bench.py --steps 3 --warmup 1 --mode infer --no-cpu-baseline > $O/pmc_a.log 2>&1
echo pmc a done
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_TRANS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_SALU --kernel-trace --output-format csv -d $O/pmc_b -- python3 bench.py --steps 3 --warmup 1 --mode infer --no-cpu-baseline > $O/pmc_b.log 2>&1
echo pmc b done
python3 scripts/pmc_mfma.py $O/pmc_mfma.json $O/kernel_stats_infer.csv $O/pmc_a $O/pmc_b
find $O -name "*.csv" -size +2M -delete
rm -rf $O/stats_infer
hipcc --offload-arch=gfx950 -O3 -Wno-unused-value scripts/microbench_valu.hip -o /tmp/mb_valu && timeout -k 10 120 /tmp/mb_valu > $O/mb_valu.log 2>&1; tail -40 $O/mb_valu.log
timeout -k 10 200 python3 scripts/ab_block.py base=transformerupscaler_amd/csrc/build/ab_base.so frag=transformerupscaler_amd/csrc/build/ab_frag.so > $O/ab_frag.log 2>&1; tail -8 $O/ab_frag.log
echo all done
