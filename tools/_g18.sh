O=gpurun_out/r03v; mkdir -p $O
export NGSAMG_NO_BUILD=1
AMGX_SETUP_LOG=1 timeout -k 10 500 python bench.py --config cfg5 --smoother gs --steps 30 --warmup 5 --no-cpu-baseline --no-reference-defaults > $O/cfg5_gs.json 2> $O/cfg5_gs.log
AMGX_SETUP_LOG=1 timeout -k 10 300 python bench.py --config cfg3 --smoother gs --steps 30 --warmup 5 --no-cpu-baseline --no-reference-defaults > $O/cfg3_gs.json 2> $O/cfg3_gs.log
AMGX_SETUP_LOG=1 timeout -k 10 300 python bench.py --config cfg3 --steps 30 --warmup 5 --no-cpu-baseline --no-reference-defaults > $O/cfg3_j.json 2> $O/cfg3_j.log
timeout -k 10 300 python bench.py --steps 100 --warmup 10 --cpu-seconds 5 --no-reference-defaults > $O/cfg2_j.json 2> $O/cfg2_j.log
grep -o '"value": [0-9.]*' $O/*.json
grep "\[bench\] assembly" $O/*.log
