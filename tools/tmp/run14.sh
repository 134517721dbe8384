set -e
for rep in 1 2; do
  for sp in 1 0; do
    LICOS_HOST_SPLIT=$sp timeout -k 10 300 python bench.py --no-extras --no-cpu-baseline --steps 6 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('split=$sp', d['value'], d['ms_per_step'])"
  done
done
for sp in 1 0; do
  LICOS_HOST_SPLIT=$sp timeout -k 10 300 taskset -c 0-2 python bench.py --no-extras --no-cpu-baseline --steps 6 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('3 cores for 16 threads, split=$sp', d['value'], d['ms_per_step'])"
done
timeout -k 10 600 python -m pytest tests/test_gpu_fp16.py tests/test_gpu_parity.py -x -q -m gpu -k "placement or coder or compress or golden or chunk or packed or model or scale" > gpurun_out/t14.log 2>&1 || { tail -60 gpurun_out/t14.log; exit 1; }
tail -3 gpurun_out/t14.log
