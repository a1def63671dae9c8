set -e
OUT=gpurun_out/r03c; mkdir -p $OUT
python -c 'import __graft_entry__ as g; g.build()' > $OUT/build.log 2>&1
export NGSAMG_NO_BUILD=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python bench.py --no-cpu-baseline --steps 200 > $OUT/bench_kt.json 2> $OUT/kt.log
f=$(find $OUT/kt -name "*kernel_stats.csv" | head -1); cp $f $OUT/kernel_stats_jacobi.csv
rm -rf $OUT/kt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt2 -- python bench.py --no-cpu-baseline --steps 200 --nv 108 > $OUT/bench_kt108.json 2> $OUT/kt108.log
f=$(find $OUT/kt2 -name "*kernel_stats.csv" | head -1); cp $f $OUT/kernel_stats_jacobi108.csv
rm -rf $OUT/kt2
