set -e
OUT=gpurun_out/r03o; mkdir -p $OUT
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $OUT/gpu_tests.log 2>&1 || { grep -v "^  File\|^Extension" $OUT/gpu_tests.log | tail -60; exit 1; }
tail -3 $OUT/gpu_tests.log
python -c 'import __graft_entry__ as g; g.smoke()' > $OUT/smoke.log 2>&1; tail -3 $OUT/smoke.log
