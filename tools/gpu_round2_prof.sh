set -o pipefail
bash tools/profile.sh r2_512 512 > gpurun_out/prof_r2_512.log 2>&1; echo "512 rc $?"
bash tools/profile.sh r2_1024 1024 > gpurun_out/prof_r2_1024.log 2>&1; echo "1024 rc $?"
ls gpurun_out/prof_r2_512 gpurun_out/prof_r2_1024
