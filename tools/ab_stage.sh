#!/bin/bash
# Dev tool: per-stage timing (tools/stage_bench.py) for A/B builds and switches.
#   tools/ab_stage.sh OUTDIR TILES
out=$1; tiles=${2:-2048}
mkdir -p $out
for cfg in "asm 1" "asm 0" "builtin 1" "builtin 0"; do
  set -- $cfg
  so=""; [ "$1" = builtin ] && so=$PWD/build/ab/liblicos_builtin.so
  echo "== glds=$1 deconv8=$2" >> $out/stages.log
  LICOS_HIP_SO=$so LICOS_DECONV8=$2 timeout -k 10 300 python tools/stage_bench.py $tiles >> $out/stages.log 2>&1 || exit 1
done
