O=gpurun_out/r03t; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export NGSAMG_NO_BUILD=1
NGSAMG_FORCE_DIST=1 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_d1 -- python bench.py --nv 108 --steps 5 --warmup 3 --no-cpu-baseline --no-reference-defaults > /dev/null 2> $O/trace_d1.log
f=$(find $O/trace_d1 -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python tools/trace_gaps.py $f 2 60,1000 25 > $O/trace_dist_world1_nv108.txt
rm -rf $O/trace_d1
AMGX_DIST_GRAPH=0 NGSAMG_FORCE_DIST=1 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_d2 -- python bench.py --nv 108 --steps 5 --warmup 3 --no-cpu-baseline --no-reference-defaults > /dev/null 2> $O/trace_d2.log
f=$(find $O/trace_d2 -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python tools/trace_gaps.py $f 2 60,1000 25 > $O/trace_dist_world1_nv108_direct.txt
rm -rf $O/trace_d2
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_p -- python bench.py --nv 108 --steps 5 --warmup 3 --no-cpu-baseline --no-reference-defaults > /dev/null 2> $O/trace_p.log
f=$(find $O/trace_p -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python tools/trace_gaps.py $f 2 60,1000 25 > $O/trace_plain_nv108.txt
rm -rf $O/trace_p
wc -l $O/*.txt
