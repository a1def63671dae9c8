O=gpurun_out/r03q; mkdir -p $O
export NGSAMG_NO_BUILD=1
NGSAMG_SETUP_LOG=1 AMGX_SETUP_LOG=1 timeout -k 10 300 python bench.py --steps 50 --warmup 5 --smoother gs --no-cpu-baseline > $O/cfg2_gs.json 2> $O/cfg2_gs.log
timeout -k 10 400 python -m pytest tests/test_gpu_devbuild.py -x -q 2>&1 | tail -3
grep -o '"value": [0-9.]*' $O/cfg2_gs.json
