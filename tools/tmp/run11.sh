set -e
timeout -k 10 300 python tools/b1024_probe.py 2>&1 | grep -v amdgpu.ids | head -8
timeout -k 10 900 python -m pytest tests/test_gpu_fp16.py tests/test_gpu_parity.py -x -q -m gpu -k "placement or coder or compress or golden or chunk or packed or model" > gpurun_out/t11.log 2>&1 || { tail -60 gpurun_out/t11.log; exit 1; }
tail -3 gpurun_out/t11.log
