"""bench.py's multi-rank launcher logic without GPUs: the watchdog that ends a run when a rank dies or nothing comes back."""
import importlib.util
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _proc(code, capture):
    return subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE if capture else subprocess.DEVNULL, text=True)


def test_wait_ranks_collects_the_result_line():
    b = _bench()
    procs = [_proc("print('noise'); print(chr(123) + 'value: 1' + chr(125))", True), _proc("import time; time.sleep(0.3)", False)]
    codes, out, why = b._wait_ranks(procs, 30.0)
    assert codes == [0, 0] and why is None and "{value: 1}" in out


def test_wait_ranks_stops_the_others_when_a_rank_fails():
    b = _bench()
    t0 = time.time()
    procs = [_proc("import time; time.sleep(60)", True), _proc("import sys; sys.exit(3)", False), _proc("import time; time.sleep(60)", False)]
    codes, out, why = b._wait_ranks(procs, 30.0)
    assert time.time() - t0 < 20 and why and "failed" in why
    assert codes[1] == 3 and codes[0] != 0 and codes[2] != 0


def test_wait_ranks_times_out():
    b = _bench()
    t0 = time.time()
    procs = [_proc("import time; time.sleep(60)", True), _proc("import time; time.sleep(60)", False)]
    codes, out, why = b._wait_ranks(procs, 1.0)
    assert time.time() - t0 < 20 and why and "no result" in why and all(c != 0 for c in codes)


def test_cpu_baseline_of_the_multi_rank_line():
    """the cpu_baseline object of an N > 1 bench line (rank 0: CPU oracle on the single-process hierarchy of the shared matrix),
    at a toy size: fields of the contract, strong / weak scaling of the value, the size guard"""
    import types
    b = _bench()
    args = types.SimpleNamespace(config="cfg2", smoother="jacobi", cpu_seconds=0.2, no_cpu_baseline=False)
    s = b._dist_cpu_baseline(args, 12, 2, True, False, {"spw": 1})
    assert s["kind"] == "port" and s["unit"] == "applies/s" and s["cores"] >= 1 and s["value"] > 0
    assert "12^3" in s["sample"] and "1728 DOF" in s["sample"]
    sp = s["single_process_hierarchy"]
    assert sp["level_sizes"][0] == 1728 and sp["operator_complexity"] > 1.0 and 0 < sp["pcg_iterations"] < 60
    w = b._dist_cpu_baseline(args, 12, 2, False, False, {"spw": 1})
    assert "one rank's 12^3 box" in w["sample"]
    args.config, args.smoother = "cfg3", "gs"
    e = b._dist_cpu_baseline(args, 6, 2, True, True, {"spw": 1})
    assert e["value"] > 0 and "648 DOF" in e["sample"]
    assert b._dist_cpu_baseline(args, 400, 8, True, True, {"spw": 1}) is None
