"""The C-ABI library loads without a GPU and exports every symbol include/dflow.h declares; parameter validation and
error reporting work on the host.  No compute calls here (CPU only)."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT, pkg


@pytest.fixture(scope="module")
def L():
    lib = pkg("_lib")
    if not os.path.exists(lib.LIB_PATH):
        lib.build()
    return lib


def test_exports_every_declared_symbol(L):
    header = open(os.path.join(ROOT, "include", "dflow.h")).read()
    declared = sorted(set(re.findall(r"\b(dflow_[a-z_]+)\s*\(", header)))
    assert "dflow_daisy" in declared and "dflow_bcd_prepare" in declared and len(declared) >= 12
    handle = C.CDLL(L.LIB_PATH)
    for name in declared:
        assert hasattr(handle, name), "libdflow.so does not export %s" % name
    assert set(L.SYMBOLS) == set(declared), "python binding and header disagree"


def test_version_and_struct_layout(L):
    assert L.lib().dflow_version() == 1
    assert C.sizeof(L.Params) == 72                       # struct dflow_params
    p = L.default_params(375, 1241, 25, 73)               # reference constants, daisy i flann.py:34-48,88,172,207-208
    assert (p.pich, p.picw, p.cellh, p.cellw) == (375, 1241, 25, 73)
    assert (p.maxnprop, p.knn, p.window, p.ngauss, p.tpsi) == (150, 5, 2, 25, 8)
    assert abs(p.tphi - 2.5) < 1e-7 and abs(p.sigma - 8.0) < 1e-7 and p.lamda == 0.05 and p.label_pitch == 160
    assert p.flags == 0 and L.FLAG_KNN_EXACT == 1          # per-call switches live in the params, not in the environment
    header = open(os.path.join(ROOT, "include", "dflow.h")).read()
    assert "getenv" not in open(os.path.join(ROOT, pkg().__name__, "csrc", "abi.hip")).read()
    assert re.search(r"#define\s+DFLOW_FLAG_KNN_EXACT\s+1\b", header)


def test_workspace_size_and_validation(L):
    lib = L.lib()
    p = L.default_params(436, 1024, 27, 64)
    ws = lib.dflow_workspace_bytes(C.byref(p))
    assert ws > 436 * 1024 * 2 * 160 * 20                 # at least the BCD bit matrices
    for field, bad, msg in (("knn", 4, b"knn"), ("window", 3, b"window"), ("maxnprop", 200, b"maxnprop"),
                            ("label_pitch", 150, b"label_pitch"), ("cellh", 0, b"cell"), ("pich", 4, b"image size"),
                            ("tpsi", 0, b"tpsi"), ("ngauss", 100, b"ngauss"), ("flags", 64, b"flags"), ("flags", 2, b"flags")):
        q = L.default_params(436, 1024, 27, 64)
        setattr(q, field, bad)
        assert lib.dflow_workspace_bytes(C.byref(q)) == 0
        assert msg in lib.dflow_last_error(), (field, lib.dflow_last_error())


def test_null_pointers_are_rejected_before_any_launch(L):
    lib = L.lib()
    p = L.default_params(64, 64, 8, 8)
    rc = lib.dflow_daisy(C.byref(p), None, None, None, 0, None)
    assert rc == -1 and b"NULL" in lib.dflow_last_error()
    rc = lib.dflow_bcd_phase(C.byref(p), 1, 1, 1, 7, 1, 1 << 40, None)
    assert rc == -1 and b"phase" in lib.dflow_last_error()
    rc = lib.dflow_bcd_sweep(C.byref(p), 1, 1, 1, 1, 16, None)
    assert rc == -2 and b"workspace" in lib.dflow_last_error()


def test_batched_entry_points_validate_on_the_host(L):
    """dflow_bcd_*_batch take host arrays of device pointers: count, phase, NULL entries and the workspace size are
    checked before anything is launched."""
    lib = L.lib()
    p = L.default_params(64, 64, 8, 8)
    ws = lib.dflow_workspace_bytes(C.byref(p))
    arr = (C.c_void_p * 2)(1, 1)
    bad = (C.c_void_p * 2)(1, None)
    assert lib.dflow_bcd_phase_batch(C.byref(p), 0, arr, arr, 0, arr, ws, None) == -1 and b"npass" in lib.dflow_last_error()
    assert lib.dflow_bcd_phase_batch(C.byref(p), 2, arr, arr, 4, arr, ws, None) == -1 and b"phase" in lib.dflow_last_error()
    assert lib.dflow_bcd_phase_batch(C.byref(p), 2, arr, bad, 0, arr, ws, None) == -1 and b"NULL" in lib.dflow_last_error()
    assert lib.dflow_bcd_sweep_batch(C.byref(p), 2, arr, arr, arr, 16, None) == -2 and b"workspace" in lib.dflow_last_error()
    assert lib.dflow_bcd_phase_batch(C.byref(p), 2, None, arr, 0, arr, ws, None) == -1


def test_knn_measurement_aids_validate_before_touching_the_gpu(L):
    """dflow_knn_screen_stats / dflow_knn_proposals_timed describe the MFMA-screened search: with DFLOW_FLAG_KNN_EXACT, a null
    output or a workspace that is too small they fail on the host."""
    lib = L.lib()
    p = L.default_params(64, 64, 8, 8)
    out = (C.c_int64 * 13)()
    assert lib.dflow_knn_screen_stats(C.byref(p), 1, 16, None, out) == -2 and b"workspace" in lib.dflow_last_error()
    assert lib.dflow_knn_screen_stats(C.byref(p), 1, 1 << 40, None, None) == -1 and b"NULL" in lib.dflow_last_error()
    q = L.default_params(64, 64, 8, 8, flags=L.FLAG_KNN_EXACT)
    assert lib.dflow_knn_screen_stats(C.byref(q), 1, 1 << 40, None, out) == -1 and b"MFMA" in lib.dflow_last_error()
    # the header's count of statistics is what the host mirror names
    import re
    header = open(os.path.join(ROOT, "include", "dflow.h")).read()
    n = int(re.search(r"#define\s+DFLOW_KNN_STATS_N\s+(\d+)", header).group(1))
    assert n == 13 and n == len(pkg("pipeline").DiscreteFlow.KNN_STATS)
    # the workspace holds the event lists: one entry per tile of the largest cell + 1 per lane (no list can run out)
    big = L.default_params(436, 1024, 27, 64)
    tiles = -(-(31 * 64) // 192) * 6                        # the last cell row is 31 px high; cells are padded to 192 rows
    nl = 16 * 16 * 31 * 25
    assert lib.dflow_workspace_bytes(C.byref(big)) > nl * 2 * (tiles + 1) * 256


def test_no_cpu_fallback(L):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(L.DflowError):
        pkg("pipeline").DiscreteFlow(64, 64, 8, 8)
