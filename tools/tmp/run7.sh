set -e
timeout -k 10 1000 bash tools/profile_round.sh gpurun_out/prof_r04 > gpurun_out/prof_r04.log 2>&1 || { tail -30 gpurun_out/prof_r04.log; ls gpurun_out/prof_r04; exit 1; }
ls -la gpurun_out/prof_r04
