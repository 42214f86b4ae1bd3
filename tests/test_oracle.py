"""CPU tests: the oracle against the committed golden vectors (which come from the reference
Fortran, tests/golden/make_golden.py) and, when oracle/_ref was built, against the reference itself."""
import os

import numpy as np
import pytest

import cases
from oracle.oracle import REF_SO, PortOracle, RefOracle


def load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    return {k: z[k] for k in z.files}


def relmax(a, b):
    m = b != 0
    return float(np.max(np.abs(a[m] - b[m]) / np.abs(b[m]))) if m.any() else 0.0


@pytest.mark.parametrize("name", list(cases.CASES))
def test_port_vs_golden(port_oracle, golden_dir, name):
    g = load(golden_dir, name)
    c = cases.CASES[name]()
    # the generator must reproduce the stored inputs exactly (same seeds, same numpy)
    for k in ("ndens", "xh", "pos", "flux"):
        assert np.array_equal(c[k], g[k]), k
    np.testing.assert_allclose(c["thick"], g["thick"], rtol=1e-12)
    c.update(thin=g["thin"], thick=g["thick"], heat_thin=g["heat_thin"], heat_thick=g["heat_thick"])
    o = cases.run_oracle(port_oracle, c)
    assert np.array_equal(o["coldensh_out"], g["coldensh_last"])          # no libm log10 involved: bit-exact
    assert np.array_equal(o["phi_ion"] != 0, g["phi_ion"] != 0)
    assert relmax(o["phi_ion"], g["phi_ion"]) <= 1e-12                    # libm log10 variants only
    if bool(g["heat_valid"]):
        assert relmax(o["phi_heat"], g["phi_heat"]) <= 1e-12
    o0 = cases.run_oracle(port_oracle, c, flux_mode=0)                    # reference bug mode (last flux for all)
    assert relmax(o0["phi_ion"], g["phi_ion_lastflux_bug"]) <= 1e-12


def test_cinterp_kat(port_oracle, golden_dir):
    z = np.load(os.path.join(golden_dir, "cinterp_kat.npz"))
    dens = np.random.default_rng(seed=42).random((20, 20, 20), dtype=np.float64)
    for pos0 in [(5, 5, 5), (1, 2, 3), (10, 5, 15)]:
        exp = z["%02d%02d%02d" % pos0]
        for i in range(4):
            for j in range(4):
                for k in range(4):
                    if i == j == k == 0:
                        continue
                    got = port_oracle.cinterp((pos0[0] + i, pos0[1] + j, pos0[2] + k), pos0, dens, 1.0)
                    assert tuple(exp[i, j, k]) == got


def test_cinterp_float32_sqrt_signature(port_oracle):
    """the body-diagonal neighbour carries sqrt(3.0) in single precision."""
    cd = np.ones((8, 8, 8))
    v, path = port_oracle.cinterp((4, 4, 4), (3, 3, 3), cd, 1e-30)
    assert v == float(np.float32(3.0) ** np.float32(0.5)) and v != np.sqrt(3.0)
    assert path == np.sqrt(3.0)


def test_last_flux_bug_is_reproduced(port_oracle):
    """with flux_mode=0 two sources (flux 1, 3) == both sources with flux 3."""
    c = cases.case_rcut7p3()
    a = cases.run_oracle(port_oracle, dict(c, flux=np.array([1.0, 3.0])), flux_mode=0)["phi_ion"]
    b = cases.run_oracle(port_oracle, dict(c, flux=np.array([3.0, 3.0])), flux_mode=1)["phi_ion"]
    assert np.array_equal(a, b)


def test_subbox_early_exit_and_photon_loss(port_oracle):
    """USE_SUBBOX semantics (raytracing.f90:193-221): a generous loss_fraction stops after one subbox."""
    c = cases.case_uniform32_center()
    kw = dict(heat_thin=None, heat_thick=None, num_tau=c["num_tau"])
    full = port_oracle.do_all_sources(c["flux"], c["pos"], c["ndens"], c["xh"], c["sig"], c["dr"], c["thin"], c["thick"],
                                      c["minlogtau"], c["dlogtau"], c["R"], max_subbox=1000, subboxsize=4,
                                      loss_fraction=0.0, **kw)
    early = port_oracle.do_all_sources(c["flux"], c["pos"], c["ndens"], c["xh"], c["sig"], c["dr"], c["thin"], c["thick"],
                                       c["minlogtau"], c["dlogtau"], c["R"], max_subbox=1000, subboxsize=4,
                                       loss_fraction=0.999999, **kw)
    assert full["nsubbox"] == 4 and early["nsubbox"] == 1
    assert full["visits"] == 32 ** 3 and early["visits"] == 9 ** 3
    assert early["photon_loss"] > full["photon_loss"] >= 0
    sl = (slice(12, 21),) * 3
    assert np.array_equal(full["phi_ion"][sl], early["phi_ion"][sl])   # growing boxes re-use finished cells


def test_visit_count(port_oracle):
    assert port_oracle.asora_visit_count(60, 256) == (1521729, 904089)
    assert port_oracle.asora_visit_count(1e3, 128)[0] == 128 ** 3


@pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref not built (needs /root/reference + amdflang)")
def test_port_vs_compiled_reference(port_oracle):
    ref = RefOracle()
    c = cases.case_multi16_unequal()
    c = dict(c, pos=c["pos"][:4], flux=c["flux"][:4])
    a, b = cases.run_oracle(ref, c), cases.run_oracle(port_oracle, c)
    assert np.array_equal(a["coldensh_out"], b["coldensh_out"])
    assert relmax(b["phi_ion"], a["phi_ion"]) <= 1e-13
    a0, b0 = cases.run_oracle(ref, c, flux_mode=0), cases.run_oracle(port_oracle, c, flux_mode=0)
    assert relmax(b0["phi_ion"], a0["phi_ion"]) <= 1e-13


@pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref not built (needs /root/reference + amdflang)")
def test_port_vs_compiled_reference_bench_config(port_oracle):
    """The cube sweep every full-size GPU comparison uses (BASELINE configs[1]: 256^3, R = 60, max_subbox = 60,
    raytracing.f90:174-175) on clustered sources in the log-normal grid: port == compiled reference."""
    rng = np.random.default_rng(918)
    N, ns = 256, 2
    nd = cases.lognormal_density(N, 7, smooth=2.0)
    p = (nd ** 2).ravel()
    idx = np.random.default_rng(11).choice(p.size, size=ns, p=p / p.sum())
    pos = np.stack(np.unravel_index(idx, (N, N, N)), axis=1).astype(np.int32)
    c = cases._base(N, 60.0, 3.0 * cases.MPC / 200, pos, 4.09e40 * rng.uniform(1e9, 5e11, ns) / 1e48, ndens=nd,
                    xh=np.full((N, N, N), 2e-4), table_kw=dict(temp=1e5, numtau=20000))
    o = cases.run_reference(port_oracle, RefOracle(), c, max_subbox=60, subboxsize=60)  # asserts port == reference
    assert o["kind"] == "reference" and (o["phi_ion"] > 0).sum() > 100_000


@pytest.mark.parametrize("name", list(cases.CASES))
def test_device_log10_variant_differs_by_log10_rounding_only(port_oracle, devlog_oracle, name):
    """The port with the HIP kernels' log10 in its lookups (oracle/log10_device.cpp) against the port with glibc's:
    identical column densities (no log10 there), identical support, and every rate differs -- beyond 1e-15 of
    itself -- by at most what a +-1 ulp change of the log10 results explains (`phi_sens`, the conditioning term of tests/parity.py, here with
    factor 2: two correct log10's differ by at most one ulp per lookup, and a rate holds two lookups).  This is the
    CPU half of the demonstration that the cells tests/parity.py admits through that term are log10 rounding
    and nothing else; the GPU half is tests/test_gpu_parity.py::test_phi_vs_device_log10_oracle."""
    from parity import record
    c = cases.CASES[name]()
    a = cases.run_oracle(port_oracle, c, sens=True)
    b = cases.run_oracle(devlog_oracle, c)
    assert np.array_equal(a["coldensh_out"], b["coldensh_out"])
    for k, s in (("phi_ion", "phi_sens"), ("phi_heat", "heat_sens")):
        assert np.array_equal(a[k] != 0, b[k] != 0)
        err = np.abs(a[k] - b[k])
        m = err > 0
        assert np.all(a[s][m] > 0), "a rate differs in a cell whose log10 sensitivity is zero"
        # (beyond a few ulps of the rate itself: where |log10 tau| << 1 one ulp of it is far below the 2^-58 absolute
        # accuracy both log10's have, `sens` is then ~0 and so is the difference -- measured <= 2.1e-16 relative)
        m = err > 1e-15 * np.abs(a[k])
        ratio = float(np.max(err[m] / a[s][m])) if m.any() else 0.0
        rel = relmax(b[k], a[k])
        print(f"[{name}] {k}: cells that differ {int(m.sum())}, max |d|/sens {ratio:.3f}, max rel {rel:.2e}, "
              f"rel > 1e-10 in {int((err > 1e-10 * np.abs(a[k])).sum())}")
        assert ratio <= 2.0
        if k == "phi_ion":
            record(f"cpu: port(glibc log10) vs port(device log10) {name}",
                   dict(max_rel=rel, n_rel_gt=int((err > 1e-10 * np.abs(a[k])).sum()), frac_floor=0.0, n_sens=0,
                        n_bad=int((err > 2.0 * a[s]).sum()), support_equal=True, ok=bool(ratio <= 2.0)),
                   max_err_over_sens=ratio, oracle="port(device log10) vs port(glibc log10)")
