#!/bin/bash
# Profiles of one round on the GPU box (gpurun): kernel-trace statistics and HBM traffic counters of bench.py.
#   tools/profile_round.sh <out_dir under gpurun_out> <commit>
# PMC passes are separate runs with --pmc only (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass).
set -o pipefail
OUT=gpurun_out/$1; COMMIT=$2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# every native library is built HERE, outside the profiler: under rocprofv3 (--pmc above all) the preloaded profiler library
# has initialised the GPU before python starts, and a compiler child spawned from the profiled process would be an exec hop
python -c 'import __graft_entry__ as g; g.build()' > $OUT/build.log 2>&1 || { echo "build failed" >> $OUT/progress.txt; exit 1; }
export NGSAMG_NO_BUILD=1
# 1. the driver's command under the kernel trace: the average of the dominant kernel must agree with roofline.kernel_ms
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_default -- python bench.py > $OUT/bench_under_rocprof_jacobi.json 2> $OUT/kt_default.log
f=$(find $OUT/kt_default -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats_jacobi.csv
rm -rf $OUT/kt_default; echo "default done" >> $OUT/progress.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_gs -- python bench.py --smoother gs --steps 50 --warmup 10 --no-cpu-baseline > $OUT/bench_under_rocprof_gs.json 2> $OUT/kt_gs.log
f=$(find $OUT/kt_gs -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats_gs.csv
rm -rf $OUT/kt_gs; echo "gs done" >> $OUT/progress.txt
# 2. counters
for sm in jacobi gs; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_${sm}_$c -- python bench.py --smoother $sm --no-graph --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2> $OUT/pmc_${sm}_$c.log
    f=$(find $OUT/pmc_${sm}_$c -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python tools/pmc_summary.py $f $OUT/pmc_${sm}_${c}_by_kernel.csv
    rm -rf $OUT/pmc_${sm}_$c
  done
  echo "pmc $sm done" >> $OUT/progress.txt
done
# 3. plain bench lines (no profiler)
python bench.py > $OUT/bench_jacobi.json 2> $OUT/bench_jacobi.err; echo "bench jacobi" >> $OUT/progress.txt
python bench.py --smoother gs --steps 100 > $OUT/bench_gs.json 2> $OUT/bench_gs.err; echo "bench gs" >> $OUT/progress.txt
python bench.py --smoother gs_mc --steps 50 --no-cpu-baseline > $OUT/bench_gs_mc.json 2> /dev/null
python bench.py --config cfg3 --steps 50 --warmup 10 --cpu-seconds 8 > $OUT/bench_cfg3.json 2> $OUT/bench_cfg3.err; echo "bench cfg3" >> $OUT/progress.txt
python bench.py --config cfg5 --steps 30 --warmup 5 --cpu-seconds 8 > $OUT/bench_cfg5.json 2> $OUT/bench_cfg5.err; echo "bench cfg5" >> $OUT/progress.txt
NGSAMG_FORCE_DIST=1 python bench.py --steps 100 --no-cpu-baseline > $OUT/bench_dist_world1.json 2> $OUT/bench_dist_world1.err
NGSAMG_FORCE_DIST=1 python bench.py --smoother gs --steps 50 --no-cpu-baseline > $OUT/bench_dist_world1_gs.json 2> /dev/null
NGSAMG_FORCE_DIST=1 python bench.py --config cfg3 --steps 30 --warmup 5 > $OUT/bench_dist_world1_cfg3.json 2> /dev/null; echo "bench dist" >> $OUT/progress.txt
# 4. timeline of the rank-partitioned cycle at world size 1, two virtual ranks at full box size
NGSAMG_FORCE_DIST=1 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_d1 -- python bench.py --steps 5 --warmup 3 --no-cpu-baseline > /dev/null 2> $OUT/trace_d1.log
f=$(find $OUT/trace_d1 -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python tools/trace_gaps.py $f 2 > $OUT/trace_dist_world1.txt
rm -rf $OUT/trace_d1
python tests/dist_fullsize_check.py 2 215x215x215 nocheck > $OUT/dist_fullsize_2x215.txt 2>&1
AMGX_DIST_NO_OVERLAP=1 python tests/dist_fullsize_check.py 2 215x215x215 nocheck 2>&1 | tail -1 >> $OUT/dist_fullsize_2x215.txt
echo $COMMIT > $OUT/commit.txt
ls $OUT
