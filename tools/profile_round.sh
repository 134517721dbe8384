#!/bin/bash
# Round profiling recipe (run on the GPU box through gpurun): kernel stats, HBM traffic (two PMC passes), MFMA-busy.
#   tools/profile_round.sh OUTDIR
set -u
out=$(realpath -m $1); mkdir -p $out
root=$(cd $(dirname $0)/.. && pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $root/bench.py --no-cpu-baseline --no-extras --steps 4 --warmup 2 > $out/bench_under_rocprof.json 2> $out/stats.err || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $out/pmc_$c -- python3 $root/bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 --batch 2048 --chunk 2048 > $out/pmc_$c.json 2> $out/pmc_$c.err || exit 1
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_mfma -- python3 $root/tools/stage_bench.py 1024 > $out/pmc_mfma.log 2>&1 || exit 1
cd $root
f=$(find $out/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1); w=$(find $out/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
python tools/pmc_traffic.py $f $w "deconv5x5s2_mfma8_kernel" 2048 5242880 $out/pmc_traffic_deconv_s4.json
python tools/pmc_traffic.py $f $w "conv5x5s2_mfma8_kernel" 2048 5242880 $out/pmc_traffic_conv_a3.json
python tools/pmc_mfma.py $(find $out/pmc_mfma -name "*counter_collection.csv" | head -1) $out/pmc_mfma_busy.json
cp $(find $out/stats -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
