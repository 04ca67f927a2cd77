timeout -k 10 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "train or grads or dp or parity or per_sample or adam" > gpurun_out/r2_t2.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_t2.log; grep -E "passed|failed|rc=|^E " gpurun_out/r2_t2.log | head -30
timeout -k 10 300 python bench.py --mode train --steps 10 --warmup 3 > gpurun_out/r2_train1.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r2_train1.json')); print('train', d['value'], d['ms_per_step'], d['loss'])"
TUP_NO_COMPOSED_TRAIN=1 timeout -k 10 300 python bench.py --mode train --steps 10 --warmup 3 > gpurun_out/r2_train0.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r2_train0.json')); print('train explicit', d['value'], d['ms_per_step'], d['loss'])"
