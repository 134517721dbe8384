set -e
timeout -k 10 300 python tools/host_coder_bench.py 256 16 5 > gpurun_out/host_coder_bench.log 2>&1 || true
cat gpurun_out/host_coder_bench.log | tail -3
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fp16.py tests/test_gpu_eval.py -x -q -m gpu > gpurun_out/t1.log 2>&1 || { tail -40 gpurun_out/t1.log; exit 1; }
tail -2 gpurun_out/t1.log
timeout -k 10 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/bench_split.json 2> gpurun_out/bench_split.err || { tail -20 gpurun_out/bench_split.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_split.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'])
print(json.dumps(d.get('batches')))
print(json.dumps(d['grid']['decode_from_plain_bytes']))
PY
