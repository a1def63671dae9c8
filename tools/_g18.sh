O=gpurun_out/r03y; mkdir -p $O
export NGSAMG_NO_BUILD=1
for rep in 1 2; do
AMGX_FUSED_EPT_MAX=4 timeout -k 10 400 python bench.py --hierarchy spw --steps 100 --no-cpu-baseline > $O/ab_ept4_$rep.json 2> /dev/null
timeout -k 10 400 python bench.py --hierarchy spw --steps 100 --no-cpu-baseline > $O/ab_ept6_$rep.json 2> /dev/null
done
AMGX_FUSED_EPT_MAX=4 timeout -k 10 400 python bench.py --hierarchy spw --smoother gs --steps 100 --no-cpu-baseline > $O/ab_gs_ept4.json 2> /dev/null
timeout -k 10 400 python bench.py --hierarchy spw --smoother gs --steps 100 --no-cpu-baseline > $O/ab_gs_ept6.json 2> /dev/null
grep -o '"value": [0-9.]*' $O/ab_*.json
