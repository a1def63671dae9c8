#!/bin/bash
# Profiles of round 4 on the GPU box (gpurun): kernel-trace statistics and HBM traffic counters of bench.py on the library-default
# (= reference) hierarchy -- cfg 2 Jacobi and Gauss-Seidel, cfg 3 / cfg 5 Gauss-Seidel -- plus the bench lines themselves.
#   tools/profile_round4.sh <out_dir under gpurun_out> <commit> [part]      part: a (cfg 2), b (cfg 3 / cfg 5), c (rank-partitioned), d (edge-matrix hierarchies, N > 1 line at world 1), default abc
# PMC passes are separate runs with --pmc only (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass).
set -o pipefail
OUT=gpurun_out/$1; COMMIT=$2; PART=${3:-abc}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# every native library is built HERE, outside the profiler: under rocprofv3 the preloaded profiler library has initialised the
# GPU before python starts, and a compiler child spawned from the profiled process would be an exec hop the pool refuses
python -c 'import __graft_entry__ as g; g.build()' > $OUT/build.log 2>&1 || { echo "build failed" >> $OUT/progress.txt; exit 1; }
export NGSAMG_NO_BUILD=1
( while true; do date >> $OUT/heartbeat.txt; sleep 60; done ) &
HB=$!
trap "kill $HB 2>/dev/null" EXIT
kt() {   # kt <tag> <bench args...>: kernel trace + stats of one bench command; stats of the launches after amgx_create beside rocprof's own
  tag=$1; shift
  timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$tag -- python bench.py "$@" > $OUT/bench_under_rocprof_$tag.json 2> $OUT/kt_$tag.log
  f=$(find $OUT/kt_$tag -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats_$tag.csv
  f=$(find $OUT/kt_$tag -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python tools/stats_after_setup.py $f $OUT/kernel_stats_${tag}_after_setup.csv
  [ -n "$f" ] && python tools/trace_gaps.py $f 2 600,20000 300 > $OUT/trace_$tag.txt 2>/dev/null
  rm -rf $OUT/kt_$tag; echo "kt $tag done" >> $OUT/progress.txt
}
pmc() {  # pmc <tag> <bench args...>: the two counter passes, summarised per kernel
  tag=$1; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    # (AMGX_NO_DENSE_TAIL: forming the collapsed coarse operator is thousands of tiny launches, minutes under the counter collection;
    #  the level-0 / level-1 kernels the counters are read for are the same either way)
    AMGX_NO_DENSE_TAIL=1 timeout -k 10 900 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_${tag}_$c -- python bench.py "$@" --no-graph --steps 5 --warmup 2 --no-cpu-baseline --no-continuity > /dev/null 2> $OUT/pmc_${tag}_$c.log
    f=$(find $OUT/pmc_${tag}_$c -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python tools/pmc_summary.py $f $OUT/pmc_${tag}_${c}_by_kernel.csv
    rm -rf $OUT/pmc_${tag}_$c
  done
  echo "pmc $tag done" >> $OUT/progress.txt
}
if [[ $PART == *a* ]]; then
  kt jacobi_spw --no-continuity          # the driver's command without the second (continuity) handle, whose kernels carry the same names
  kt gs_spw --smoother gs --steps 50 --warmup 10 --no-cpu-baseline --no-continuity
  pmc jacobi_spw
  pmc gs_spw --smoother gs
  python bench.py > $OUT/bench_jacobi.json 2> $OUT/bench_jacobi.err; echo "bench jacobi" >> $OUT/progress.txt
  python bench.py --smoother gs --steps 100 --no-continuity > $OUT/bench_gs.json 2> $OUT/bench_gs.err; echo "bench gs" >> $OUT/progress.txt
  python bench.py --ops --steps 100 --no-cpu-baseline --no-continuity > /dev/null 2> $OUT/ops_jacobi.txt
  python bench.py --smoother gs --ops --steps 50 --no-cpu-baseline --no-continuity > /dev/null 2> $OUT/ops_gs.txt
  python bench.py --nv 108 --no-cpu-baseline --no-continuity > $OUT/bench_nv108.json 2> /dev/null
fi
if [[ $PART == *b* ]]; then
  for cfg in cfg3 cfg5; do
    kt ${cfg}_gs_spw --config $cfg --smoother gs --steps 30 --warmup 5 --no-cpu-baseline --no-continuity
    pmc ${cfg}_gs_spw --config $cfg --smoother gs
    python bench.py --config $cfg --smoother gs --steps 30 --warmup 5 --cpu-seconds 6 > $OUT/bench_${cfg}_gs.json 2> $OUT/bench_${cfg}_gs.err
    python bench.py --config $cfg --steps 30 --warmup 5 --cpu-seconds 6 --no-continuity > $OUT/bench_$cfg.json 2> $OUT/bench_$cfg.err
    python bench.py --config $cfg --smoother gs --hierarchy aaf --steps 30 --warmup 5 --cpu-seconds 6 > $OUT/bench_${cfg}_gs_aaf.json 2> $OUT/bench_${cfg}_gs_aaf.err
    echo "bench $cfg" >> $OUT/progress.txt
  done
fi
if [[ $PART == *c* ]]; then
  NGSAMG_FORCE_DIST=1 python bench.py --steps 100 --no-cpu-baseline > $OUT/bench_dist_world1.json 2> $OUT/bench_dist_world1.err
  NGSAMG_FORCE_DIST=1 python bench.py --nv 108 --steps 200 --no-cpu-baseline > $OUT/bench_dist_world1_nv108.json 2> /dev/null
  NGSAMG_FORCE_DIST=1 python bench.py --smoother gs --steps 50 --no-cpu-baseline > $OUT/bench_dist_world1_gs.json 2> /dev/null
  NGSAMG_FORCE_DIST=1 python bench.py --config cfg5 --smoother gs --steps 30 --warmup 5 > $OUT/bench_dist_world1_cfg5_gs.json 2> /dev/null; echo "bench dist" >> $OUT/progress.txt
fi
if [[ $PART == *d* ]]; then      # part d (not in the default set): what round 4 left unmeasured on the GPU
  # the edge-matrix hierarchies at full size (general blocks in P: no rigid-body transfer storage), with sp_improve_its beside them
  for cfg in cfg3 cfg5; do
    python bench.py --config $cfg --smoother gs --edge-mats 1 --steps 30 --warmup 5 --cpu-seconds 6 --no-continuity > $OUT/bench_${cfg}_gs_edge_mats.json 2> $OUT/bench_${cfg}_gs_edge_mats.err
    echo "bench $cfg edge_mats" >> $OUT/progress.txt
  done
  python bench.py --config cfg3 --smoother gs --edge-mats 2 --steps 30 --warmup 5 --cpu-seconds 6 --no-continuity > $OUT/bench_cfg3_gs_edge_mats_robust.json 2> $OUT/bench_cfg3_gs_edge_mats_robust.err
  # the N > 1 bench line's code path (cpu_baseline of the shared matrix, single-process hierarchy) at world size 1
  NGSAMG_FORCE_DIST=1 python bench.py --nv 108 --steps 200 --cpu-seconds 5 > $OUT/bench_dist_world1_nv108_with_cpu_baseline.json 2> $OUT/bench_dist_world1_nv108_with_cpu_baseline.err
  NGSAMG_SETUP_LOG=1 python bench.py --steps 20 --no-cpu-baseline --no-continuity > /dev/null 2> $OUT/setup_log_cfg2_two_pass_contract.txt
fi
echo $COMMIT > $OUT/commit_part_$PART.txt
ls $OUT
