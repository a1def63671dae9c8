O=gpurun_out/r03w; mkdir -p $O
export NGSAMG_NO_BUILD=1
for nv in 171 136 108; do
  NGSAMG_FORCE_DIST=1 timeout -k 10 300 python bench.py --nv $nv --steps 200 --warmup 20 --no-cpu-baseline --no-reference-defaults > $O/dist_world1_nv$nv.json 2> $O/dist_world1_nv$nv.log
  timeout -k 10 300 python bench.py --nv $nv --steps 200 --warmup 20 --no-cpu-baseline --no-reference-defaults > $O/plain_nv$nv.json 2> $O/plain_nv$nv.log
done
timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-reference-defaults > $O/plain_nv215.json 2> $O/plain_nv215.log
AMGX_SETUP_LOG=1 timeout -k 10 500 python bench.py --config cfg5 --smoother gs --steps 30 --warmup 5 --no-cpu-baseline --no-reference-defaults > $O/cfg5_gs.json 2> $O/cfg5_gs.log
grep -o '"value": [0-9.]*' $O/*.json
grep "\[bench\] assembly" $O/cfg5_gs.log
