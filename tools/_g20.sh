O=gpurun_out/r03z; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export NGSAMG_NO_BUILD=1
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python bench.py --hierarchy spw --steps 50 --warmup 10 --no-cpu-baseline --no-reference-defaults > $O/bench_spw_rocprof.json 2> $O/kt.log
f=$(find $O/kt -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python tools/stats_after_setup.py $f $O/kernel_stats_jacobi_spw_after_setup.csv
[ -n "$f" ] && python tools/trace_gaps.py $f 2 500,5000 100 > $O/trace_spw.txt
rm -rf $O/kt
head -20 $O/kernel_stats_jacobi_spw_after_setup.csv | cut -c1-80,250-
cat $O/trace_spw.txt | cut -c1-130
