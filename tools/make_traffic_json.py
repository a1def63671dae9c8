#!/usr/bin/env python3
"""profiles/traffic_*.json from the per-kernel PMC summaries of tools/profile_round.sh:
HBM-side bytes per launch = 2 * FETCH_SIZE + WRITE_SIZE (KB -> bytes; gfx950 correction of MI355X_MICROARCH.md, HBM section).
python tools/make_traffic_json.py <dir with pmc_*_by_kernel.csv> <commit> <date>"""
import csv
import json
import os
import sys


def mean(path, needle):
    """the (kernel, grid) group with the LARGEST mean among those matching: level 0 of the kernel"""
    best = (None, 0)
    if not os.path.exists(path):
        return best
    for r in csv.DictReader(open(path)):
        if needle in r["kernel"] and (best[0] is None or float(r["mean_per_launch"]) > best[0]):
            best = (float(r["mean_per_launch"]), int(r["launches"]))
    return best


def main():
    d, commit, date = sys.argv[1], sys.argv[2], sys.argv[3]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    jobs = [("jacobi", "sell_pre_restrict_kernel<512, 0", "traffic_pre_restrict_l0.json", "sell_pre_restrict_kernel<512> level 0 (cfg 2)"),
            ("jacobi", "sell_win_spmv_kernel<512, 2>", "traffic_q_l0.json", "sell_win_spmv_kernel<512, EP_AXPY> on Q, level 0 (cfg 2)"),
            ("gs", "sell_spmv_kernel<1, 1>", "traffic_spmv_l0.json", "sell_spmv_kernel<1, EP_RES> level 0 (cfg 2)"),
            ("gs", "gsb_sweep_kernel<256, 1, false", "traffic_gsb_sweep_l0.json", "gsb_sweep_kernel<256, 1, false, 8> level 0 (cfg 2)"),
            ("gs", "sell_win_cres_restrict_kernel<512", "traffic_gs_res_restrict_l0.json", "sell_win_cres_restrict_kernel<512> level 0 (cfg 2)"),
            ("cfg3_jacobi", "bsell_spmv_kernel<3, 1>", "traffic_bsell_res_cfg3.json", "bsell_spmv_kernel<3, EP_RES> level 0 (cfg 3: r = b - A x, 3x3 blocks)"),
            ("cfg5_jacobi", "bsell_spmv_kernel<6, 1>", "traffic_bsell_res_cfg5.json", "bsell_spmv_kernel<6, EP_RES> level 0 (cfg 5: r = b - A x, 6x6 blocks)"),
            ("cfg5_jacobi", "bsell_spmv_kernel<6, 3>", "traffic_bsell_jac_cfg5.json", "bsell_spmv_kernel<6, EP_JAC> level 0 (cfg 5: folded block-Jacobi pre-smoothing pass)"),
            ("cfg3_gs", "bgsb_sweep_kernel<3, false>", "traffic_bgsb_sweep_cfg3.json", "bgsb_sweep_kernel<3, false> level 0 (cfg 3: backward block-hybrid Gauss-Seidel sweep)"),
            ("cfg5_gs", "bgsb_sweep_kernel<6, false>", "traffic_bgsb_sweep_cfg5.json", "bgsb_sweep_kernel<6, false> level 0 (cfg 5: backward block-hybrid Gauss-Seidel sweep)"),
            ("cfg5_gs", "bsell_spmv_kernel<6, 0>", "traffic_bgsb_rest_cfg5.json", "bsell_spmv_kernel<6, EP_MULT> on the rest copy, level 0 (cfg 5: residual after the sweep from zero)"),
            ("cfg5_gs", "rb_prolong_kernel<6, 6, 3", "traffic_rb_prolong_cfg5.json", "rb_prolong_kernel<6, 6, 3> level 0 (cfg 5: x + P x_c with rigid-body blocks)"),
            ("cfg5_gs", "rb_restrict_kernel<6, 6, 3", "traffic_rb_restrict_cfg5.json", "rb_restrict_kernel<6, 6, 3> level 0 (cfg 5: P^T r with rigid-body blocks)")]
    for sm, needle, out, label in jobs:
        f, nf = mean(os.path.join(d, f"pmc_{sm}_FETCH_SIZE_by_kernel.csv"), needle)
        w, nw = mean(os.path.join(d, f"pmc_{sm}_WRITE_SIZE_by_kernel.csv"), needle)
        if f is None:
            continue
        if nf == 0:
            continue
        w = w or 0.0
        js = {"kernel": label, "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "launches_averaged": [nf, nw],
              "correction": "gfx950: FETCH_SIZE counts 1/2 of the bytes of coalesced streaming reads (MI355X_MICROARCH.md, HBM section; "
                            "calibrated in round 1: profiles/r01/pmc_lab_calibration.csv); WRITE_SIZE exact",
              "hbm_bytes_per_launch": int(round((2.0 * f + w) * 1024)),
              "note": "L2<->fabric bytes; separate rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of `bench.py --no-graph --steps 5 --warmup 2 "
                      "--no-cpu-baseline` (tools/profile_round3.sh; AMGX_NO_DENSE_TAIL=1 for the counter passes), averaged over the launches of the kernel on its largest grid (level 0)",
              "commit": commit, "collected": date}
        json.dump(js, open(os.path.join(root, "profiles", out), "w"), indent=1)
        print(out, js["hbm_bytes_per_launch"])


if __name__ == "__main__":
    main()
