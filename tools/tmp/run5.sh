set -e
timeout -k 10 600 python -m pytest tests/test_gpu_fp16.py -x -q -m gpu -k "many_bands or 13_bands" > gpurun_out/t5.log 2>&1 || { tail -60 gpurun_out/t5.log; exit 1; }
tail -3 gpurun_out/t5.log
for f in 1 0 1 0; do
  LICOS_FIRST16=$f timeout -k 10 200 python tools/stage_bench.py 2048 13 2>&1 | grep -E "conv_13|total" | sed "s/^/first16=$f  /" >> gpurun_out/first16_ab.log
done
cat gpurun_out/first16_ab.log
for f in 1 0; do
  LICOS_FIRST16=$f timeout -k 10 300 python tools/hyper_probe.py 2048 2048 2>&1 | grep -E "iter 2|conv_13" | sed "s/^/first16=$f  /" >> gpurun_out/first16_hyper.log
done
cat gpurun_out/first16_hyper.log
timeout -k 10 900 python -m pytest tests/test_gpu_hyperprior.py tests/test_gpu_fp16.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/t5b.log 2>&1 || { tail -60 gpurun_out/t5b.log; exit 1; }
tail -3 gpurun_out/t5b.log
