"""CPU tests of the evolve-loop host logic and of the oracle's chemistry against the goldens
generated from the compiled reference (tests/golden/make_golden.py evolve)."""
import os

import numpy as np
import pytest

import cases
from evolve_host import evolve3D_host
from pyc2ray_amd.evolve import _evolve_loop, conv_criterion


def test_conv_criterion_formula():
    # pyc2ray/evolve.py:154
    assert conv_criterion(1e-4, 128 ** 3, 1) == 0
    assert conv_criterion(1e-4, 128 ** 3, 10) == 3
    assert conv_criterion(1e-4, 32 ** 3, 1000) == 3
    assert conv_criterion(1e-4, 250 ** 3, 100000) == 1562


def test_loop_stops_on_either_criterion():
    # (a) few non-converged cells  (b) both relative changes small
    seq = iter([(500, 10.0, 90.0), (100, 20.0, 80.0), (2, 30.0, 70.0)])
    n = _evolve_loop(lambda: None, lambda: next(seq), 1000, 100, 1e-2)  # crit = min(10, 33)
    assert n == 3
    seq = iter([(500, 10.0, 90.0), (500, 10.0005, 89.9995), (500, 0, 0)])
    hist = []
    n = _evolve_loop(lambda: None, lambda: next(seq), 1000, 100, 1e-2, history=hist)
    assert n == 2 and hist[1]["rel_change_xh1"] < 1e-2
    # zero sums count as "changed" (evolve.py:353-361)
    seq = iter([(500, 0.0, 100.0), (500, 0.0, 100.0), (0, 0.0, 100.0)])
    assert _evolve_loop(lambda: None, lambda: next(seq), 1000, 100, 1e-2) == 3


def test_loop_decision_is_collective():
    calls = []

    def decide(flag):
        calls.append(flag)
        return len(calls) >= 3  # "rank 0" says converged only at the third iteration

    n = _evolve_loop(lambda: None, lambda: (0, 1.0, 1.0), 1000, 100, 1e-2, decide=decide)
    assert n == 3 and calls == [True, True, True]


def test_port_evolve_vs_reference_golden(port_oracle, golden_dir):
    g = np.load(os.path.join(golden_dir, "evolve32.npz"))
    r = evolve3D_host(port_oracle, cases.evolve_case())
    assert r["niter"] == int(g["niter"])
    assert r["conv_flags"] == list(g["conv_flags"])
    assert np.max(np.abs(r["xh_new"] - g["xh_new"])) <= 1e-13
    assert np.max(np.abs(r["xh_av"] - g["xh_av"])) <= 1e-13
    m = g["phi_ion"] > 0
    assert np.max(np.abs(r["phi_ion"][m] / g["phi_ion"][m] - 1.0)) <= 1e-10


def test_port_chemistry_vs_reference_golden(port_oracle, golden_dir):
    g = np.load(os.path.join(golden_dir, "chemistry24.npz"))
    c = cases.chemistry_case()
    xa, xi = c["xh_av"].copy(), c["xh_intermed"].copy()
    cf = port_oracle.global_pass(c["dt"], c["ndens"], c["temp"], c["xh"], xa, xi, c["phi_ion"], c["clump"], c["bh00"],
                                 c["albpow"], c["colh0"], c["temph0"], c["abu_c"])
    assert cf == int(g["conv_flag"])
    assert np.max(np.abs(xa - g["xh_av"])) <= 1e-15 and np.max(np.abs(xi - g["xh_intermed"])) <= 1e-15


def test_global_pass_argument_validation():
    from pyc2ray_amd import libasora
    c = cases.chemistry_case(N=4)
    args = [c[k] for k in ("ndens", "temp", "xh", "xh_av", "xh_intermed", "phi_ion", "clump")]
    chem = [c[k] for k in ("bh00", "albpow", "colh0", "temph0", "abu_c")]
    with pytest.raises(TypeError):
        libasora.global_pass(c["dt"], args[0].astype(np.float32), *args[1:], *chem)
    with pytest.raises(ValueError):
        libasora.global_pass(c["dt"], np.asfortranarray(args[0]), *args[1:], *chem)
    with pytest.raises(RuntimeError):  # no device initialised in the CPU suite
        libasora.global_pass(c["dt"], *args, *chem)
