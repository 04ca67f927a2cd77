timeout -k 10 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "kernels or hip_model or parity or hip_rt or window" > gpurun_out/r2_t3.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2_t3.log
grep -E "passed|failed|rc=|^E " gpurun_out/r2_t3.log | head -30
grep -q "rc=0" gpurun_out/r2_t3.log && timeout -k 10 200 python bench.py --steps 30 --warmup 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['ms_per_launch'], d['roofline']['frac'])"
