O=gpurun_out/r03q; mkdir -p $O
export NGSAMG_NO_BUILD=1
timeout -k 10 500 python -m pytest tests/test_gpu_devbuild.py -x -q -k "rank_partitioned" > $O/devbuild.log 2>&1; echo "rc=$?" >> $O/devbuild.log
tail -25 $O/devbuild.log
