set -e
build/lat > gpurun_out/lat.log 2>&1 || true
tail -30 gpurun_out/lat.log
timeout -k 10 200 python tools/power_probe.py 4096 10 > gpurun_out/power_probe.log 2>&1 || true
tail -12 gpurun_out/power_probe.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format rocpd -d $GRAFT_REPO_ROOT/gpurun_out/htrain -o t -- python3 $GRAFT_REPO_ROOT/tools/train_bench.py 16 13 10 0 bmshj2018-hyperprior > $GRAFT_REPO_ROOT/gpurun_out/htrain_bench.log 2>&1 || true
cd $GRAFT_REPO_ROOT
python3 tools/kernel_summary.py $(find gpurun_out/htrain -name "t_results.db" | head -1) gpurun_out/htrain_kernel_geometry.csv 100 || true
rm -rf gpurun_out/htrain
tail -3 gpurun_out/htrain_bench.log
head -25 gpurun_out/htrain_kernel_geometry.csv
timeout -k 10 900 python bench.py > gpurun_out/bench_r04a.json 2> gpurun_out/bench_r04a.err || { tail -20 gpurun_out/bench_r04a.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_r04a.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['frac_conv_only'])
print(json.dumps(d.get('batches')))
g=d['grid']['configs']
for k in ('1ch','13ch'):
    print(k, {kk:vv for kk,vv in g[k].items() if kk!='weights'})
h=g['hyperprior_13x512']
for k in ('B2048','B4096','B256'):
    if k in h: print(k, {kk:vv for kk,vv in h[k].items() if kk not in ('stages',)})
print(h.get('quality_match'))
print('fp32', d['grid'].get('fp32_path_B16384'))
print('train', d['grid'].get('train_step'))
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['quality_match'])
PY
