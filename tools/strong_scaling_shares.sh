#!/bin/bash
# per-rank compute shares of the cfg-2 strong-scaling run, measured on ONE GPU (see profiles/r04/strong_scaling_shares.json)
#   tools/strong_scaling_shares.sh <out_dir under gpurun_out>
O=gpurun_out/$1; mkdir -p $O
export NGSAMG_NO_BUILD=1
( while true; do date >> $O/heartbeat.txt; sleep 60; done ) &
HB=$!
trap "kill $HB 2>/dev/null" EXIT
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-continuity > $O/plain_215.json 2> /dev/null
NGSAMG_FORCE_DIST=1 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/dist_215.json 2> /dev/null
for nv in 171 136 108; do
  python bench.py --nv $nv --steps 400 --warmup 40 --no-cpu-baseline --no-continuity > $O/plain_$nv.json 2> /dev/null
  NGSAMG_FORCE_DIST=1 python bench.py --nv $nv --steps 400 --warmup 40 --no-cpu-baseline > $O/dist_$nv.json 2> /dev/null
  echo "nv $nv done" >> $O/progress.txt
done
python - <<PY
import json
O = "$O"
t1 = json.load(open(f"{O}/plain_215.json"))["ms_per_step"]
d1 = json.load(open(f"{O}/dist_215.json"))["ms_per_step"]
out = {"what": "per-rank compute share of the strong-scaling run of cfg 2 (215^3, library-default = reference hierarchy), measured on ONE GPU: the "
               "rank-partitioned cycle through RCCL at world size 1 on a grid with the rows one of N ranks owns (171^3 ~ 1/2, 136^3 ~ 1/4, 108^3 ~ 1/8 "
               "of 9.94 M) and the plain single-GPU handle on the same grid; the wire (ncclSend/Recv of halo planes, ncclAllGather of the gathered "
               "level) comes on top at N > 1.  efficiency_bound = t1 / (N * dist_world1)",
       "t1_ms": t1, "t1_dist_world1_ms": d1, "shares": []}
for N, nv in ((2, 171), (4, 136), (8, 108)):
    p = json.load(open(f"{O}/plain_{nv}.json")); d = json.load(open(f"{O}/dist_{nv}.json"))
    out["shares"].append({"N": N, "nv": nv, "rows": nv ** 3, "dist_world1_ms": d["ms_per_step"], "plain_ms": p["ms_per_step"],
                          "speedup_bound": round(t1 / d["ms_per_step"], 2), "efficiency_bound": round(t1 / (N * d["ms_per_step"]), 3),
                          "efficiency_bound_plain_handle": round(t1 / (N * p["ms_per_step"]), 3)})
json.dump(out, open(f"{O}/strong_scaling_shares.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
