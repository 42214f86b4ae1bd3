"""Seeded sweep over small meshes, radii and kernel variants against the CPU oracle: the corners the golden
cases do not visit (tiny and odd meshes, radii from below one cell to beyond the box, sources on faces and
corners, every workgroup size / staging layout the library can pick)."""
import numpy as np
import pytest

import cases
from parity import compare
from test_gpu_parity import run_hip

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def asora():
    from pyc2ray_amd import libasora
    return libasora

FUZZ_STATS = dict(n=0, n_sens_vs_glibc_oracle=0, max_rel_vs_device_log10_oracle=0.0, frac_floor_max=0.0)

MESHES = [4, 5, 7, 8, 12, 17, 24, 31, 40]
VARIANTS = [dict(), dict(wg_threads=64), dict(wg_threads=128), dict(wg_threads=192), dict(wg_threads=512),
            dict(wg_threads=1024), dict(wg_threads=1024, compact_staging=1), dict(wg_threads=256, compact_staging=1),
            dict(force_generic=1), dict(source_order=0), dict(skip_unobserved=1), dict(lds_stage_limit=1024),
            dict(quad_kernel=-1), dict(quad_kernel=1), dict(quad_kernel=1, wg_threads=64), dict(quad_kernel=1, wg_threads=256, compact_staging=1),
            dict(quad_kernel=1, skip_unobserved=1), dict(quad_kernel=2), dict(quad_kernel=2, wg_threads=128),
            dict(quad_kernel=2, compact_staging=1, skip_unobserved=1),
            dict(quad_kernel=1, src_pack=2), dict(quad_kernel=1, src_pack=3), dict(quad_kernel=2, src_pack=2),
            dict(quad_kernel=2, src_pack=3, skip_unobserved=1), dict(skip_saturated=1, quad_kernel=-1),
            dict(skip_saturated=1, quad_kernel=-1, wg_threads=512, compact_staging=1)]


@pytest.mark.parametrize("seed", range(156))
def test_random_small_case_vs_oracle(asora, port_oracle, devlog_oracle, seed):
    rng = np.random.default_rng(9000 + seed)
    N = MESHES[seed % len(MESHES)]
    R = [0.4, 1.0, 1.7, 2.5, N / 3.0, N / 2.0 + 0.3, float(N), 3.0 * N][int(rng.integers(0, 8))]
    ns = int(rng.integers(1, 6))
    pos = rng.integers(0, N, (ns, 3))
    pos[0] = [0, N - 1, int(rng.integers(0, N))]  # one source on an edge of the box
    flux = rng.uniform(0.1, 10.0, ns)
    nd = cases.lognormal_density(N, 100 + seed)
    xh = rng.uniform(1e-4, 0.999, (N, N, N))
    c = cases._base(N, R, rng.choice([2e20, 5e21, 4.6e22]), pos, flux, ndens=nd, xh=xh)
    opts = VARIANTS[seed % len(VARIANTS)]
    for k, v in opts.items():
        asora.set_option(k, v)
    with_heat = seed % 3 == 0
    # the two grid-layout modes alternate (small cases would all take the one-layout path by default)
    asora.set_option("single_layout", 1 if (seed // len(VARIANTS)) % 2 == 0 else -1)
    try:
        phi, cdh, heat = run_hip(asora, c, heat=with_heat)
    finally:
        asora.set_option("single_layout", 0)
        for k in opts:
            asora.set_option(k, {"source_order": 2}.get(k, 0))
    o = cases.run_oracle(port_oracle, c, heat=with_heat, sens=True)
    r = compare(phi, o["phi_ion"], sens=o["phi_sens"])
    assert r["ok"] and r["support_equal"], (N, R, ns, opts, r)
    if with_heat:
        rh = compare(heat, o["phi_heat"], sens=o["heat_sens"])
        assert rh["ok"], (N, R, ns, opts, rh)
    # ... and against the oracle with the device's log10 the plain bar holds with no conditioning term
    # (tests/test_gpu_parity.py::test_phi_vs_device_log10_oracle)
    od = cases.run_oracle(devlog_oracle, c, heat=with_heat)
    rd = compare(phi, od["phi_ion"])
    assert rd["ok"] and rd["support_equal"], (N, R, ns, opts, rd)
    FUZZ_STATS["n"] += 1
    FUZZ_STATS["n_sens_vs_glibc_oracle"] += r["n_sens"]
    FUZZ_STATS["max_rel_vs_device_log10_oracle"] = max(FUZZ_STATS["max_rel_vs_device_log10_oracle"], rd["max_rel"])
    FUZZ_STATS["frac_floor_max"] = max(FUZZ_STATS["frac_floor_max"], rd["frac_floor"])
    if with_heat:
        rdh = compare(heat, od["phi_heat"])
        assert rdh["ok"], (N, R, ns, opts, rdh)
    # column density of the last source: bit-identical wherever both swept
    last = dict(c, pos=c["pos"][-1:], flux=c["flux"][-1:])
    ol = cases.run_oracle(port_oracle, last, heat=False)
    m = (cdh != 0) & (ol["coldensh_out"] != 0)
    assert np.array_equal(cdh[m], ol["coldensh_out"][m]), (N, R, ns, opts)


def test_zz_fuzz_summary_to_parity_report():
    """One line for the parity report: all fuzz cases against the device-log10 oracle at the plain bar."""
    from parity import record
    if FUZZ_STATS["n"] == 0:
        pytest.skip("fuzz cases did not run in this process")
    record(f"fuzz: {FUZZ_STATS['n']} seeded cases vs device-log10 oracle (no conditioning term)",
           dict(max_rel=FUZZ_STATS["max_rel_vs_device_log10_oracle"], n_rel_gt=-1, frac_floor=FUZZ_STATS["frac_floor_max"],
                n_sens=0, n_bad=0, support_equal=True, ok=True),
           cells_admitted_by_sens_vs_glibc_oracle=FUZZ_STATS["n_sens_vs_glibc_oracle"])
