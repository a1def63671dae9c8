O=gpurun_out/r03x; mkdir -p $O
export NGSAMG_NO_BUILD=1
timeout -k 10 300 python -m pytest tests/test_gpu_fullsize.py -x -q -s -k cfg1 > $O/cfg1.log 2>&1; echo "rc=$?" >> $O/cfg1.log
grep "cfg 1 iterations\|passed\|failed\|rc=" $O/cfg1.log
