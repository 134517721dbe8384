set -e
timeout -k 10 900 python -m pytest tests/test_gpu_fp16.py tests/test_gpu_parity.py -x -q -m gpu -k "placement or coder or compress or golden or chunk or packed or model" > gpurun_out/t9.log 2>&1 || { tail -60 gpurun_out/t9.log; exit 1; }
tail -3 gpurun_out/t9.log
timeout -k 10 900 python bench.py --no-cpu-baseline > gpurun_out/bench_r04c.json 2> gpurun_out/bench_r04c.err || { tail -20 gpurun_out/bench_r04c.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_r04c.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'])
print(json.dumps(d.get('batches')))
print(d['grid']['fp32_path'])
PY
