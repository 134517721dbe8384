set -e
timeout -k 10 900 python -m pytest tests/test_gpu_fp16.py -x -q -m gpu > gpurun_out/t8.log 2>&1 || { tail -60 gpurun_out/t8.log; exit 1; }
tail -3 gpurun_out/t8.log
timeout -k 10 900 python bench.py > gpurun_out/bench_r04b.json 2> gpurun_out/bench_r04b.err || { tail -20 gpurun_out/bench_r04b.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_r04b.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['frac_conv_only'], d['roofline_g_a2']['frac'])
print(json.dumps(d.get('batches')))
g=d['grid']['configs']
for k in ('1ch','13ch'):
    print(k, {kk:vv for kk,vv in g[k].items() if kk!='weights'})
h=g['hyperprior_13x512']
for k in ('B2048','B4096','B256'):
    if k in h: print(k, {kk:vv for kk,vv in h[k].items() if kk not in ('stages',)})
print('train', d['grid'].get('train_step'), d['grid'].get('train_step_hyperprior'))
print('fp32', d['grid'].get('fp32_path_B16384'))
print('cpu', d['cpu_baseline']['value'])
print(len(json.dumps(d)))
PY
