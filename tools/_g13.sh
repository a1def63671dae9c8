set -e
OUT=gpurun_out/r03m; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_hgs.py tests/test_gpu_dense_tail.py tests/test_gpu_edge_cases.py -x -q > $OUT/tests.log 2>&1 || { grep -v "^  File\|^Extension" $OUT/tests.log | tail -60; exit 1; }
tail -3 $OUT/tests.log
python bench.py --config cfg5 --smoother gs --steps 50 --warmup 10 --no-cpu-baseline --no-reference-defaults --ops > $OUT/bench_cfg5_gs_compact.json 2> $OUT/bench_cfg5_gs_compact.log
AMGX_BGSB_LINE_BLOCKS=1 python bench.py --config cfg5 --smoother gs --steps 50 --warmup 10 --no-cpu-baseline --no-reference-defaults > $OUT/bench_cfg5_gs_lines.json 2> $OUT/bench_cfg5_gs_lines.log
python bench.py --config cfg3 --smoother gs --steps 50 --warmup 10 --no-cpu-baseline --no-reference-defaults > $OUT/bench_cfg3_gs_compact.json 2> $OUT/bench_cfg3_gs_compact.log
grep -H -o '"value": [0-9.]*' $OUT/*.json
grep -h "cycle step\|upload" $OUT/bench_cfg5_gs_compact.log | cut -c1-140
python bench.py --config cfg3 --smoother gs --steps 50 --warmup 10 --cpu-seconds 4 --no-reference-defaults > $OUT/bench_cfg3_gs_full.json 2> $OUT/bench_cfg3_gs_full.log
grep -o '"gs_iterations": {[^}]*}' $OUT/bench_cfg3_gs_full.json
