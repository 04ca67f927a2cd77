timeout -k 10 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "kernels or model or parity" > gpurun_out/r2_t2.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_t2.log; grep -E "passed|failed|rc=|^E " gpurun_out/r2_t2.log | head -30
timeout -k 10 200 python scripts/microbench_block.py 1920 11 2>/dev/null | tail -3
