O=gpurun_out/r03q; mkdir -p $O
export NGSAMG_NO_BUILD=1
AMGX_SETUP_LOG=1 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-reference-defaults > $O/cfg2_j.json 2> $O/cfg2_j.log
AMGX_SETUP_LOG=1 AMGX_SETUP_SERIAL=1 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-reference-defaults > $O/cfg2_js.json 2> $O/cfg2_js.log
nproc
