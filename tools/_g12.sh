set -e
OUT=gpurun_out/r03_profiles_c; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_dist.py tests/test_gpu_hgs.py -x -q > $OUT/tests.log 2>&1 || { grep -v "^  File\|^Extension" $OUT/tests.log | tail -60; exit 1; }
tail -3 $OUT/tests.log
bash tools/profile_round3.sh r03_profiles_c eb234a1+wip c
