#!/usr/bin/env python3
"""profiles/traffic_*_spw.json (the library-default = reference hierarchy) from the per-kernel PMC summaries of tools/profile_round4.sh:
HBM-side bytes per launch = 2 * FETCH_SIZE + WRITE_SIZE (KB -> bytes; gfx950 correction of MI355X_MICROARCH.md, HBM section).
python tools/make_traffic_json_r04.py <dir with pmc_*_by_kernel.csv> <commit> <date>"""
import csv
import json
import os
import sys


def groups(path, needle):
    """(mean per launch, launches) of every (kernel, grid) group matching, largest mean first: level 0, level 1, ..."""
    out = []
    if not os.path.exists(path):
        return out
    for r in csv.DictReader(open(path)):
        if needle in r["kernel"]:
            out.append((float(r["mean_per_launch"]), int(r["launches"])))
    return sorted(out, reverse=True)


def coloured_sweep(path, needle):
    """block-coloured sweep of level 0: one launch per block colour, the colours have (nearly) equal grids -- the groups whose grid is
    within 10 % of the LARGEST grid of the kernel (coarser levels have smaller grids); bytes per SWEEP = sum over those launches /
    number of sweeps (= the smallest launch count among the groups: one colour)"""
    g = []
    if not os.path.exists(path):
        return None
    for r in csv.DictReader(open(path)):
        if needle in r["kernel"]:
            grid = int(r["kernel"].rsplit("[grid ", 1)[1].rstrip("]"))
            g.append((grid, float(r["mean_per_launch"]), int(r["launches"])))
    if not g:
        return None
    g0 = max(x[0] for x in g)
    lv0 = [x for x in g if x[0] >= 0.9 * g0]
    sweeps = min(x[2] for x in lv0)
    return sum(x[1] * x[2] for x in lv0) / sweeps, sweeps


def main():
    d, commit, date = sys.argv[1], sys.argv[2], sys.argv[3]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # (pmc tag, kernel-name needle, rank of the grid group by size, output, label)
    jobs = [("jacobi_spw", "sell_pre_restrict_kernel<512, 0", 0, "traffic_pre_restrict_l0_spw.json", "sell_pre_restrict_kernel<512, 0, EPT, 1> level 0 (cfg 2, default hierarchy)"),
            ("jacobi_spw", "sell_lw_win_spmv_kernel<512, 2>", 0, "traffic_q_l0_spw.json", "sell_lw_win_spmv_kernel<512, EP_AXPY> on Q, level 0 (cfg 2, default hierarchy)"),
            ("jacobi_spw", "sell_lw_pre_restrict_kernel<2, 2, 0>", 0, "traffic_lw_pre_restrict_l1_spw.json", "sell_lw_pre_restrict_kernel<2, 2, 0> level 1 (cfg 2, default hierarchy: 1.24 M rows x 52)"),
            ("jacobi_spw", "sell_lw_win_spmv_kernel<512, 2>", 1, "traffic_q_l1_spw.json", "sell_lw_win_spmv_kernel<512, EP_AXPY> on Q, level 1 (cfg 2, default hierarchy)"),
            ("jacobi_spw", "restrict_sum_kernel", 0, "traffic_restrict_sum_l0_spw.json", "restrict_sum_kernel level 0 -> 1 (cfg 2, default hierarchy)"),
            ("gs_spw", "gsb_sweep_kernel<256, 1, false", 0, "traffic_gsb_sweep_l0_spw.json", "gsb_sweep_kernel<256, 1, false, 8> level 0 (cfg 2, default hierarchy)"),
            ("gs_spw", "sell_win_cres_restrict_kernel<512", 0, "traffic_gs_res_restrict_l0_spw.json", "sell_win_cres_restrict_kernel<512> level 0 (cfg 2, default hierarchy)"),
            ("gs_spw", "sell_spmv_kernel<1, 1>", 0, "traffic_spmv_l0_spw.json", "sell_spmv_kernel<1, EP_RES> level 0 (cfg 2, default hierarchy)"),
            ("cfg3_gs_spw", "bgsb_sweep_kernel<3, 0>", 0, "traffic_bgsb_sweep_cfg3_spw.json", "bgsb_sweep_kernel<3, 0> level 0 (cfg 3: ONE block colour of the backward block-coloured sweep; a sweep = the launches of all colours)"),
            ("cfg5_gs_spw", "bgsb_sweep_kernel<6, 0>", 0, "traffic_bgsb_sweep_cfg5_spw.json", "bgsb_sweep_kernel<6, 0> level 0 (cfg 5: ONE block colour of the backward block-coloured sweep)")]
    for sm, needle, rank, out, label in jobs:
        fg = groups(os.path.join(d, f"pmc_{sm}_FETCH_SIZE_by_kernel.csv"), needle)
        wg = groups(os.path.join(d, f"pmc_{sm}_WRITE_SIZE_by_kernel.csv"), needle)
        if len(fg) <= rank:
            continue
        f, nf = fg[rank]
        w, nw = wg[rank] if len(wg) > rank else (0.0, 0)
        if "bgsb_sweep" in needle:
            cf = coloured_sweep(os.path.join(d, f"pmc_{sm}_FETCH_SIZE_by_kernel.csv"), needle)
            cw = coloured_sweep(os.path.join(d, f"pmc_{sm}_WRITE_SIZE_by_kernel.csv"), needle)
            if cf:
                f, nf = cf
                w, nw = cw if cw else (0.0, 0)
            label = label.replace("ONE block colour of the", "all block colours of the").replace("; a sweep = the launches of all colours", "")
        js = {"kernel": label, "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "launches_averaged": [nf, nw],
              "correction": "gfx950: FETCH_SIZE counts 1/2 of the bytes of coalesced streaming reads (MI355X_MICROARCH.md, HBM section; "
                            "calibrated in round 1: profiles/r01/pmc_lab_calibration.csv); WRITE_SIZE exact",
              "hbm_bytes_per_launch": int(round((2.0 * f + w) * 1024)),
              "note": "L2<->fabric bytes; separate rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of `bench.py --no-graph --steps 5 --warmup 2 "
                      "--no-cpu-baseline --no-continuity` (tools/profile_round4.sh; AMGX_NO_DENSE_TAIL=1 for the counter passes), averaged over the launches of the "
                      "kernel on the grid named in `kernel`",
              "commit": commit, "collected": date}
        json.dump(js, open(os.path.join(root, "profiles", out), "w"), indent=1)
        print(out, js["hbm_bytes_per_launch"])


if __name__ == "__main__":
    main()
