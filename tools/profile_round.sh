#!/bin/bash
# Profiles of one round on the GPU box (gpurun): kernel-trace statistics and HBM traffic counters of bench.py.
#   tools/profile_round.sh <out_dir under gpurun_out> <commit>
# PMC passes are separate runs with --pmc only (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass).
set -o pipefail
OUT=gpurun_out/$1; COMMIT=$2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for sm in jacobi gs; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$sm -- python bench.py --smoother $sm --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_under_rocprof_$sm.json 2> $OUT/kt_$sm.log
  f=$(find $OUT/kt_$sm -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats_$sm.csv
  rm -rf $OUT/kt_$sm
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_${sm}_$c -- python bench.py --smoother $sm --no-graph --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2> $OUT/pmc_${sm}_$c.log
    f=$(find $OUT/pmc_${sm}_$c -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python tools/pmc_summary.py $f $OUT/pmc_${sm}_${c}_by_kernel.csv
    rm -rf $OUT/pmc_${sm}_$c
  done
  echo "$sm done" >> $OUT/progress.txt
done
echo $COMMIT > $OUT/commit.txt
ls -la $OUT
