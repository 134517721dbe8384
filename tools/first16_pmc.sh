#!/bin/bash
# HBM-side traffic of the 13-band first stage alone (tools/first_stamps_probe.py first16 TILES): two separate rocprofv3 --pmc
# passes (FETCH_SIZE, WRITE_SIZE) per variant, reduced by tools/pmc_traffic.py.   tools/first16_pmc.sh OUTDIR [TILES]
set -u
out=$(realpath -m $1); tiles=${2:-1024}; mkdir -p $out
root=$(cd $(dirname $0)/.. && pwd)
cd /tmp && export TMPDIR=/tmp
for duo in ${DUOS:-0 1}; do
  export LICOS_FIRST16_DUO=$duo
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $out/pmc_${duo}_$c -- python3 $root/tools/first_stamps_probe.py first16 $tiles > $out/pmc_${duo}_$c.log 2>&1 || exit 1
  done
  f=$(find $out/pmc_${duo}_FETCH_SIZE -name "*counter_collection.csv" | head -1); w=$(find $out/pmc_${duo}_WRITE_SIZE -name "*counter_collection.csv" | head -1)
  k=conv5x5s2_first16_kernel; [ $duo = 1 ] && k=conv5x5s2_first16_duo_kernel
  # algorithmic bytes per 13 x 512^2 tile: 13 x 512^2 fp32 in + 128 x 256^2 fp16 out
  python3 $root/tools/pmc_traffic.py $f $w $k $tiles 30408704 $out/pmc_traffic_first16_duo$duo.json || exit 1
  rm -rf $out/pmc_${duo}_FETCH_SIZE $out/pmc_${duo}_WRITE_SIZE
done
