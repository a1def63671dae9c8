O=gpurun_out/r03x; mkdir -p $O
export NGSAMG_NO_BUILD=1
( while true; do date >> $O/heartbeat.txt; sleep 60; done ) &
HB=$!
timeout -k 10 1000 python -m pytest tests/test_gpu_dist.py tests/test_gpu_devbuild.py tests/test_gpu_hgs.py tests/test_gpu_edge_cases.py tests/test_gpu_dense_tail.py -x -q > $O/tests2.log 2>&1; echo "rc=$?" >> $O/tests2.log
kill $HB
tail -6 $O/tests2.log
