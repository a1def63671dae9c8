set -x
mkdir -p gpurun_out/r03p
export NGSAMG_NO_BUILD=1
timeout -k 10 500 python -m pytest tests/test_gpu_dist.py -q > gpurun_out/r03p/dist1.log 2>&1; echo "rc1=$?" >> gpurun_out/r03p/dist1.log
tail -5 gpurun_out/r03p/dist1.log
timeout -k 10 300 python -m pytest tests/test_gpu_dist.py -q -k "whole_cycle or allgather or pcg" > gpurun_out/r03p/dist2.log 2>&1; echo "rc2=$?" >> gpurun_out/r03p/dist2.log
tail -5 gpurun_out/r03p/dist2.log
timeout -k 10 200 NGSAMG_FORCE_DIST=1 python bench.py --nv 108 --steps 200 --warmup 20 --no-cpu-baseline --no-reference-defaults > gpurun_out/r03p/dist108.json 2> gpurun_out/r03p/dist108.log
timeout -k 10 200 python bench.py --nv 108 --steps 200 --warmup 20 --no-cpu-baseline --no-reference-defaults > gpurun_out/r03p/plain108.json 2> gpurun_out/r03p/plain108.log
timeout -k 10 300 python bench.py --steps 50 --warmup 5 --smoother gs --no-cpu-baseline --no-reference-defaults > gpurun_out/r03p/cfg2_gs.json 2> gpurun_out/r03p/cfg2_gs.log
grep -o '"value": [0-9.]*' gpurun_out/r03p/*.json
