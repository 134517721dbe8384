set -e
mkdir -p gpurun_out/weights
timeout -k 10 600 python tools/split_probe.py > gpurun_out/split_probe2.log 2>&1 || { tail -30 gpurun_out/split_probe2.log; exit 1; }
tail -42 gpurun_out/split_probe2.log
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fp16.py tests/test_gpu_eval.py -x -q -m gpu > gpurun_out/t2.log 2>&1 || { tail -40 gpurun_out/t2.log; exit 1; }
tail -2 gpurun_out/t2.log
timeout -k 10 330 python tools/train_weights.py --channels 1 --quality 5 --lmbda 0.025 --steps 100000 --minutes 3.5 --out gpurun_out/weights/factorized_q5_c1.pth.tar --log gpurun_out/r04_train_fact_q5_c1.jsonl > gpurun_out/train_c1.log 2>&1 || { tail -20 gpurun_out/train_c1.log; exit 1; }
tail -3 gpurun_out/train_c1.log
timeout -k 10 400 python tools/train_weights.py --channels 13 --quality 5 --lmbda 0.025 --steps 100000 --minutes 4.5 --out gpurun_out/weights/factorized_q5_c13.pth.tar --log gpurun_out/r04_train_fact_q5_c13.jsonl > gpurun_out/train_c13.log 2>&1 || { tail -20 gpurun_out/train_c13.log; exit 1; }
tail -3 gpurun_out/train_c13.log
