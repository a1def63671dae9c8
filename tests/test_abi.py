"""The C-ABI libraries load and export every symbol their headers declare (no compute without a GPU)."""
import os
import re

import pytest

from ngsamg_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header, prefix):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(" + prefix + r"_[a-z0-9_]+)\s*\(", txt)))


def test_amgx_symbols_exported():
    lib = _lib.hip()
    names = _declared("amgx.h", "amgx")
    assert len(names) >= 17
    for n in names:
        assert hasattr(lib, n), n
    assert set(names) == set(_lib.AMGX_SYMBOLS)


def test_amgh_symbols_exported():
    lib = _lib.host()
    names = _declared("amgh.h", "amgh")
    for n in names:
        assert hasattr(lib, n), n
    assert set(names) == set(_lib.AMGH_SYMBOLS)


def test_struct_sizes_match_headers(tmp_path):
    """layout guard: the ctypes mirrors must have the sizes the C compiler gives the header structs"""
    import ctypes as C
    import subprocess
    src = tmp_path / "sz.c"
    src.write_text(
        '#include <stdio.h>\n#include "amgx.h"\n#include "amgh.h"\n'
        'int main(void){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(amgx_matrix), sizeof(amgx_level_desc),'
        ' sizeof(amgx_hierarchy_desc), sizeof(amgh_matrix), sizeof(amgh_options), sizeof(amgh_level),'
        ' sizeof(amgx_halo_desc), sizeof(amgx_dist_desc), sizeof(amgx_gss4_desc));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)])
    sizes = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    mirrors = [_lib.amgx_matrix, _lib.amgx_level_desc, _lib.amgx_hierarchy_desc, _lib.amgh_matrix,
               _lib.amgh_options, _lib.amgh_level, _lib.amgx_halo_desc, _lib.amgx_dist_desc, _lib.amgx_gss4_desc]
    assert sizes == [C.sizeof(m) for m in mirrors]


def test_apply_path_fails_loudly_without_gpu():
    """No CPU fallback: creating the device hierarchy without a GPU must raise, not degrade."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from tests.problems import poisson_case
    from ngsamg_amd.device import DeviceAMGMatrix
    p, H = poisson_case((9, 9), "left|top", 5)
    with pytest.raises(_lib.NgsAMGError, match="no HIP device|hip"):
        DeviceAMGMatrix(H, sm_type="jacobi")


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under ngsamg_amd/, include/ or tools/ may reference it
    (only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do)."""
    bad = []
    for top in ("ngsamg_amd", "include", "tools"):
      for base, _, files in os.walk(os.path.join(ROOT, top)):
        for f in files:
            if not f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")):
                continue
            txt = open(os.path.join(base, f), errors="ignore").read()
            if re.search(r"\boracle\b", txt) and re.search(r"import\s+oracle|from\s+oracle|oracle/|liboracle", txt):
                bad.append(os.path.join(base, f))
    assert not bad, bad
