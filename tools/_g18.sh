O=gpurun_out/r03y; mkdir -p $O
export NGSAMG_NO_BUILD=1
timeout -k 10 400 python bench.py > $O/bench_default.json 2> $O/bench_default.log
timeout -k 10 400 python bench.py --hierarchy spw --steps 100 --no-cpu-baseline > $O/bench_spw.json 2> $O/bench_spw.log
timeout -k 10 400 python bench.py --hierarchy spw --smoother gs --steps 100 > $O/bench_spw_gs.json 2> $O/bench_spw_gs.log
grep -o '"value": [0-9.]*' $O/*.json | head; grep "levels:\|level [0-9]:" $O/bench_spw.log | head -12
