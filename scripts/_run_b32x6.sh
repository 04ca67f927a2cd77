timeout -k 10 1000 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "train or window or rt or dp" > gpurun_out/r2_t3.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2_t3.log
grep -E "passed|failed|rc=|^E " gpurun_out/r2_t3.log | head -30
grep -q "rc=0" gpurun_out/r2_t3.log && timeout -k 10 300 python bench.py --mode train --steps 20 --warmup 5 2>/dev/null | tail -1 | cut -c1-300 && TUP_NO_PACK_PLAN=1 timeout -k 10 300 python bench.py --mode train --steps 20 --warmup 5 2>/dev/null | tail -1 | cut -c1-200 && timeout -k 10 300 python bench.py --mode rt 2>/dev/null | tail -1 | cut -c1-200 && TUP_NO_PACK_PLAN=1 timeout -k 10 300 python bench.py --mode rt 2>/dev/null | tail -1 | cut -c1-200
