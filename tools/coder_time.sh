#!/bin/bash
# Dev tool: standalone durations of the rANS kernels (one 4096-tile chunk, nothing else on the GPU while they run).
# usage (on the GPU box, from the repo root): bash tools/coder_time.sh OUTDIR
set -e
O=$1; mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/coder -o c -- python3 $R/tools/profile_step.py 4096 4096 > $O/coder.log 2>&1
grep -E "rans_(en|de)code" $O/coder/c_kernel_stats.csv | cut -d, -f1-4 | sed 's/(.*)//' > $O/coder_summary.txt
cat $O/coder_summary.txt
