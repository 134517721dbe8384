#!/bin/bash
# Round profiling recipe (run on the GPU box through gpurun): kernel stats + per-geometry summaries of the headline step,
# the config-5 (scale hyperprior) step, the config-4 training step and the fp32 parity path; HBM traffic of the two big MFMA kernels at the
# bench's own 4096-tile launch (two separate PMC passes); MFMA-busy counters.
#   tools/profile_round.sh OUTDIR
set -u
out=$(realpath -m $1); mkdir -p $out
root=$(cd $(dirname $0)/.. && pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv rocpd -d $out/stats -o bench -- python3 $root/bench.py --no-cpu-baseline --no-extras --steps 4 --warmup 2 > $out/bench_under_rocprof.json 2> $out/stats.err || exit 1
python3 $root/tools/kernel_summary.py $(find $out/stats -name "bench_results.db" | head -1) $out/bench_kernel_geometry.csv 200 || exit 1
cp $(find $out/stats -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
rocprofv3 --kernel-trace --output-format rocpd -d $out/hyper -o h -- python3 $root/tools/hyper_probe.py 4096 2048 > $out/hyper_probe.log 2>&1 || exit 1
python3 $root/tools/kernel_summary.py $(find $out/hyper -name "h_results.db" | head -1) $out/hyper_kernel_geometry.csv 200 || exit 1
rocprofv3 --kernel-trace --output-format rocpd -d $out/train -o t -- python3 $root/tools/train_bench.py 16 13 10 0 > $out/train_bench.log 2>&1 || exit 1
python3 $root/tools/kernel_summary.py $(find $out/train -name "t_results.db" | head -1) $out/train_kernel_geometry.csv 100 || exit 1
rocprofv3 --kernel-trace --output-format rocpd -d $out/fp32 -o f -- python3 $root/tools/fp32_probe.py 16384 1024 > $out/fp32_probe.log 2>&1 || exit 1
python3 $root/tools/kernel_summary.py $(find $out/fp32 -name "f_results.db" | head -1) $out/fp32_kernel_geometry.csv 100 || exit 1
# (PMC passes with the device coding every tile: with the host's share a launch covers fewer tiles than the scaling assumes)
export LICOS_HOST_CODER=0
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $out/pmc_$c -- python3 $root/bench.py --no-cpu-baseline --no-extras --steps 1 --warmup 1 --batch 4096 --chunk 4096 > $out/pmc_$c.json 2> $out/pmc_$c.err || exit 1
done
# config 5's dominant kernel (the g_s[4]-shaped transposed conv at 128^2 -> 256^2, 2048 tiles per launch): HBM traffic
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $out/pmc_h_$c -- python3 $root/tools/hyper_probe.py 2048 2048 > $out/pmc_h_$c.log 2> $out/pmc_h_$c.err || exit 1
done
unset LICOS_HOST_CODER
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_mfma -- python3 $root/tools/stage_bench.py 1024 > $out/pmc_mfma.log 2>&1 || exit 1
cd $root
f=$(find $out/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1); w=$(find $out/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
python tools/pmc_traffic.py $f $w "deconv5x5s2_mfma8_kernel" 4096 5242880 $out/pmc_traffic_deconv_s4.json
python tools/pmc_traffic.py $f $w "conv5x5s2_mfma8_kernel" 4096 5242880 $out/pmc_traffic_conv_a3.json
python tools/pmc_traffic.py $f $w "conv5x5s2_first_raw_kernel" 4096 4980736 $out/pmc_traffic_first.json
python tools/pmc_traffic.py $f $w "deconv5x5s2_rows_kernel" 4096 4980736 $out/pmc_traffic_rows.json
fh=$(find $out/pmc_h_FETCH_SIZE -name "*counter_collection.csv" | head -1); wh=$(find $out/pmc_h_WRITE_SIZE -name "*counter_collection.csv" | head -1)
python tools/pmc_traffic.py $fh $wh "deconv5x5s2_mfma8_kernel" 2048 20971520 $out/pmc_traffic_hyper_deconv.json
python tools/pmc_traffic.py $fh $wh "conv5x5s2_first16_duo_kernel" 2048 30408704 $out/pmc_traffic_first16.json
python tools/pmc_traffic.py $fh $wh "deconv5x5s2_rows16_kernel" 2048 30408704 $out/pmc_traffic_last16.json
python tools/pmc_mfma.py $(find $out/pmc_mfma -name "*counter_collection.csv" | head -1) $out/pmc_mfma_busy.json
rm -rf $out/stats $out/hyper $out/train $out/fp32 $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE $out/pmc_mfma $out/pmc_h_FETCH_SIZE $out/pmc_h_WRITE_SIZE
