O=gpurun_out/r03q; mkdir -p $O
export NGSAMG_NO_BUILD=1
timeout -k 10 500 python -m pytest tests/test_gpu_devbuild.py -x -q > $O/devbuild.log 2>&1; echo "rc=$?" >> $O/devbuild.log
tail -5 $O/devbuild.log
