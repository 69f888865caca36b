"""CPU-side checks of the drop-in boundary: the shared library loads without a GPU, exports every
symbol include/deepaco_hip.h declares, the ctypes table covers them all, and argument validation
answers with error codes (no compute is launched here)."""
import os
import re

import pytest

from conftest import ROOT
from deepaco_amd import _lib
import oracle


def header_symbols():
    text = open(os.path.join(ROOT, "include", "deepaco_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(daco_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_something():
    syms = header_symbols()
    assert "daco_tsp_sample" in syms and "daco_pheromone_update" in syms and len(syms) >= 9


def test_library_exports_every_declared_symbol():
    L = _lib.lib()
    for s in header_symbols():
        assert hasattr(L, s), f"{s} declared in include/deepaco_hip.h but not exported"


def test_ctypes_table_matches_header():
    assert sorted(_lib.SIGNATURES) == header_symbols()


def test_version_and_layout_helpers_agree_with_oracle():
    L = _lib.lib()
    assert L.daco_version() >= 123
    for n in (2, 5, 63, 64, 65, 100, 128, 129, 255, 256, 257, 500, 1000, 4096):
        assert L.daco_vec_for_n(n) == oracle.vec_for_n(n)
        assert L.daco_ld_for_n(n) == oracle.ld_for_n(n)
        assert L.daco_ld_for_n(n) >= n and L.daco_ld_for_n(n) % (64 * L.daco_vec_for_n(n)) == 0


def test_bad_arguments_return_error_codes():
    L = _lib.lib()
    assert L.daco_tour_costs(None, 0, 0, 0, 0, None, 0, None, 0, None) == -1
    assert b"bad argument" in L.daco_last_error()
    # n above the register plan of the sampler -> DACO_E_TOOLARGE before anything is launched
    rc = L.daco_tsp_sample(None, 1, 5000, 4, 1, 0, 1, 0, 1.0, 1.0, 2, 1, None, -1, None, 0, 0, None, 0, 0, 1, None, None,
                           None, None, 0, None, None, 1, 1 << 40, None, None)
    assert rc == -2 and b"DACO_MAX_NODES" in L.daco_last_error()
    # workspace too small
    rc = L.daco_tsp_sample(None, 1, 100, 4, 1, 0, 1, 0, 1.0, 1.0, 2, 1, None, -1, None, 0, 0, None, 0, 0, 1, None, None,
                           None, None, 0, None, None, 1, 16, None, None)
    assert rc == -4
    assert L.daco_tsp_sample_workspace_bytes(64, 500, _lib.SCAN) == 64 * 500 * 512 * 4
    assert L.daco_tsp_sample_workspace_bytes(64, 500, _lib.RACE_PHILOX) == 2 * 64 * 500 * 512 * 4


def test_product_has_no_cpu_path():
    import torch
    from deepaco_amd import engine
    from deepaco_amd.tsp.aco import ACO
    d = torch.rand(5, 5)
    if torch.cuda.is_available():          # host tensors are staged to the HIP device, never computed on the CPU
        assert ACO(d, n_ants=4).distances.is_cuda
    else:
        with pytest.raises(_lib.DacoError):
            ACO(d, n_ants=4)
    with pytest.raises(_lib.DacoError):
        engine.tsp_sample(d, d, 4)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "deepaco_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", src, flags=re.M), f
                assert "daco_oracle" not in src or f.endswith(".h") and "restated in" in src, f


def test_two_opt_candidate_entry_points_validate_arguments():
    import torch
    from deepaco_amd import engine
    L = _lib.lib()
    n = 500
    al = lambda x: (x + 255) & ~255
    assert L.daco_two_opt_tables_bytes(3, n) == 3 * (256 + al(8 * n * n) + al(2 * n * n))
    assert L.daco_two_opt_tables_bytes(0, n) == 0
    assert L.daco_two_opt_prepare(None, 0, n, None, 0, None, 0) == -1
    assert L.daco_two_opt_prepare(None, 1, 2000, 1, 0, 1, 1 << 40) == -2 and b"1024" in L.daco_last_error()
    assert L.daco_two_opt_prepare(None, 1, n, 1, 0, 1, 16) == -4                      # tables buffer too small
    assert L.daco_two_opt_nbr(None, 1, 1, 2000, 1, 0, 1, 1, 1, 10, None) == -2
    assert L.daco_two_opt_nbr(None, 1, 1, n, 1, 0, None, None, 1, 10, None) == -1
    assert L.daco_two_opt_auto(None, 1, 1, n, 1, None, 0, 1, 1, 1, 10, None) == -1    # sweeps is required (hand-over state)
    # above the table kernels' size the host falls back to the dense kernel (no tables)
    assert engine.two_opt_tables(torch.zeros(1, 1025, 1025)) is None
