O=gpurun_out/r03u; mkdir -p $O
export NGSAMG_NO_BUILD=1
( while true; do date >> $O/heartbeat.txt; sleep 60; done ) &
HB=$!
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; echo "rc=$?" >> $O/gpu_tests.log
kill $HB
tail -8 $O/gpu_tests.log
