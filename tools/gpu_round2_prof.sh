set -o pipefail
bash tools/profile.sh r2_512 512 > gpurun_out/prof_r2_512.log 2>&1; echo "512 rc $?"
bash tools/profile.sh r2_1024 1024 > gpurun_out/prof_r2_1024.log 2>&1; echo "1024 rc $?"
bash tools/profile.sh r2_512_nocull 512 --no-cull > gpurun_out/prof_r2_512_nocull.log 2>&1; echo "nocull rc $?"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r2_bench_final.json 2> gpurun_out/r2_bench_final.err; echo "bench rc $?"
python tools/dropin_times.py 100 512 > gpurun_out/r2_dropin.txt 2>&1; tail -4 gpurun_out/r2_dropin.txt
