set -e
mkdir -p gpurun_out/weights
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/t_all.log 2>&1 || { tail -40 gpurun_out/t_all.log; exit 1; }
tail -3 gpurun_out/t_all.log
timeout -k 10 420 python tools/train_weights.py --channels 13 --quality 5 --lmbda 0.025 --steps 100000 --minutes 5.5 --init licos_amd/weights/factorized_q5_c13.pth.tar --out gpurun_out/weights/factorized_q5_c13.pth.tar --log gpurun_out/r04_train_fact_q5_c13_b.jsonl > gpurun_out/train_c13b.log 2>&1 || { tail -20 gpurun_out/train_c13b.log; exit 1; }
tail -3 gpurun_out/train_c13b.log
timeout -k 10 300 python tools/hyper_probe.py 2048 2048 > gpurun_out/hyper_probe.log 2>&1 || true
tail -30 gpurun_out/hyper_probe.log
