set -e
timeout -k 10 300 python -m pytest tests/test_hip_kernels.py tests/test_hip_model.py -x -q -k "tail or fused_paths" --timeout 200 2>&1 | tail -3
for i in 1 2; do
echo "== new bench"; timeout -k 10 200 python bench.py --mode infer --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'])"
echo "== old bench"; TUP_LIB_PATH=$PWD/transformerupscaler_amd/libtupscale_old.so timeout -k 10 200 python bench.py --mode infer --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'])"
done
