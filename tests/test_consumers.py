"""SURVEY 8(f) #2/#3 consumers against the reference's own code (tests/golden): the EpicFlow match exporter
(napravi_parove.py:3-13) and the error metrics of visualization.py:128-152.  CPU only."""
import numpy as np
import pytest

from conftest import GOLDEN_NAMES, pkg


@pytest.mark.parametrize("name", GOLDEN_NAMES)
def test_parovi_text_matches_reference(golden, name, tmp_path):
    g = golden(name)
    ev = pkg("evaluate")
    out = tmp_path / "parovi.txt"
    ev.parovi(g["sparse_t3"], str(out))
    assert out.read_bytes() == g["parovi_t3_txt"].tobytes()


@pytest.mark.parametrize("name", GOLDEN_NAMES)
def test_error_metrics_match_reference(golden, oracle, name):
    g = golden(name)
    ev = pkg("evaluate")
    O = oracle
    p = O.make_params(int(g["H"]), int(g["W"]), int(g["cellh"]), int(g["cellw"]), seed=int(g["seed"]))
    flow = O.full_pass(p, g["img1"], g["img2"], int(g["bcd_times"]))["flows"][-1]      # the reference's final forward flow
    gt = ev.to_uv_valid(g["gt"].astype(np.float64), g["gt_valid"])
    mean_epe, outliers, n = ev.error_metrics(ev.to_uv_valid(flow), gt)
    assert n == int(g["gt_valid"].sum())
    assert str(np.float32(mean_epe)) == str(g["epe_txt"])
    assert str(outliers) == str(g["outliers_txt"])
