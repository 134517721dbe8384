#!/bin/bash
# HBM traffic of the two end stages at the bench's 4096-tile launch: two separate rocprofv3 --pmc passes (FETCH_SIZE,
# WRITE_SIZE), reduced by tools/pmc_traffic.py.   tools/pmc_end_stages.sh OUTDIR
set -u
out=$(realpath -m $1); mkdir -p $out
root=$(cd $(dirname $0)/.. && pwd)
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $out/pmc_$c -- python3 $root/bench.py --no-cpu-baseline --no-extras --steps 1 --warmup 1 --batch 4096 --chunk 4096 > $out/pmc_$c.json 2> $out/pmc_$c.err || exit 1
done
cd $root
f=$(find $out/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1); w=$(find $out/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
# algorithmic bytes per tile: first stage 3 x 256^2 fp32 in + 128 x 128^2 fp16 out; last stage 128 x 128^2 fp16 in + 3 x 256^2 fp32 out
python tools/pmc_traffic.py $f $w "conv5x5s2_first_raw_kernel" 4096 4980736 $out/pmc_traffic_first.json
python tools/pmc_traffic.py $f $w "deconv5x5s2_rows_kernel" 4096 4980736 $out/pmc_traffic_rows.json
rm -rf $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE
