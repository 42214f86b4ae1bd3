"""GPU parity tests proper: HIP path through the C-ABI vs committed goldens and vs the CPU oracle."""
import os

import numpy as np
import pytest

import cases
from parity import compare, record

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def asora():
    from pyc2ray_amd import libasora
    return libasora


def run_hip(asora, c, heat=False, num_src=None):
    N = c["N"]
    ns = len(c["flux"]) if num_src is None else num_src
    asora.device_init(N, 8, 0, 1)
    try:
        asora.photo_table_to_device(np.ascontiguousarray(c["thin"]), np.ascontiguousarray(c["thick"]), int(c["num_tau"]))
        if heat:
            asora.heat_table_to_device(np.ascontiguousarray(c["heat_thin"]), np.ascontiguousarray(c["heat_thick"]),
                                       int(c["num_tau"]))
        asora.density_to_device(np.ravel(c["ndens"]).astype(np.float64), N)
        pos = np.ascontiguousarray(np.asarray(c["pos"], dtype=np.int32).reshape(-1, 3)[:ns].ravel())
        asora.source_data_to_device(pos, np.ascontiguousarray(c["flux"][:ns], dtype=np.float64), ns)
        cdh = np.zeros(N ** 3)
        phi = np.zeros(N ** 3)
        xh = np.ravel(c["xh"]).astype(np.float64)
        if heat:
            ph = np.zeros(N ** 3)
            asora.do_all_sources_heat(c["R"], cdh, c["sig"], c["dr"], None, xh, phi, ph, ns, N, c["minlogtau"],
                                      c["dlogtau"], int(c["num_tau"]))
        else:
            ph = None
            asora.do_all_sources(c["R"], cdh, c["sig"], c["dr"], None, xh, phi, ns, N, c["minlogtau"], c["dlogtau"],
                                 int(c["num_tau"]))
    finally:
        asora.device_close()
    sh = (N, N, N)
    return phi.reshape(sh), cdh.reshape(sh), None if ph is None else ph.reshape(sh)


def load_golden(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    c = {k: z[k] for k in z.files}
    for k in ("N", "num_tau"):
        c[k] = int(c[k])
    for k in ("R", "dr", "sig", "minlogtau", "dlogtau"):
        c[k] = float(c[k])
    return c


@pytest.fixture(params=[-1, 1], ids=["two_layouts", "one_layout"])
def layout(request, asora):
    """Both grid-layout modes of a call (option "single_layout"): the transposed copies + merge pass of long calls, and
    the one-layout path short calls take (which these small cases would all take by default)."""
    asora.set_option("single_layout", request.param)
    yield request.param
    asora.set_option("single_layout", 0)


@pytest.mark.parametrize("name", list(cases.CASES))
def test_phi_vs_golden(asora, golden_dir, name, layout):
    """phi_ion per cell vs the Fortran reference's golden vector (tolerance: see tests/parity.py)."""
    c = load_golden(golden_dir, name)
    phi, cdh, _ = run_hip(asora, c)
    assert asora.last_launch_info()["single_layout"] == (layout == 1)
    r = compare(phi, c["phi_ion"], sens=c["phi_sens"])
    record(f"golden {name} (N={c['N']}, R={c['R']:g}, {len(c['flux'])} sources, {'one layout' if layout == 1 else 'two layouts'})",
           r, oracle="reference (golden)")
    print(f"\n[{name}] max_rel={r['max_rel']:.3e} n_rel>1e-10={r['n_rel_gt']} frac_floor={r['frac_floor']:.2e} "
          f"n_sens={r['n_sens']} support_equal={r['support_equal']}")
    if name != "lognormal48":  # the one fixture built to be ill-conditioned (tests/parity.py)
        assert r["n_sens"] == 0
    assert r["ok"], r
    assert r["support_equal"], "set of cells with a non-zero rate differs (R / column-density predicates)"


@pytest.mark.parametrize("name", list(cases.CASES))
def test_phi_vs_device_log10_oracle(asora, devlog_oracle, name):
    """What the conditioning term of tests/parity.py stands for, demonstrated: against the oracle whose lookups use
    the device's own log10 (oracle/log10_device.cpp: the host build of asora_log10.hpp -- the ONE change to
    c2ray_oracle.c) the HIP rates meet the plain bar, 1e-10 relative + 1e-14 * max floor, with NO conditioning term:
    every cell, lognormal48's 543 ill-conditioned ones included, and the heating rate too.  Together with the CPU half
    (tests/test_oracle.py::test_device_log10_variant_differs_by_log10_rounding_only: glibc's and the device's log10
    move the oracle's own rates by at most 2 * sens) this shows those cells are log10 rounding and nothing else."""
    c = cases.CASES[name]()
    phi, cdh, heat = run_hip(asora, c, heat=True)
    o = cases.run_oracle(devlog_oracle, c, heat=True)
    r = compare(phi, o["phi_ion"])  # no `sens`
    rh = compare(heat, o["phi_heat"])
    record(f"device-log10 oracle {name} (N={c['N']}, R={c['R']:g}, {len(c['flux'])} sources)", r,
           oracle="port with the device's log10 (no conditioning term)", heat_max_rel=rh["max_rel"],
           heat_n_bad=rh["n_bad"], heat_frac_floor=rh["frac_floor"])
    print(f"\n[{name} vs device-log10 oracle] phi max_rel={r['max_rel']:.3e} n_rel>1e-10={r['n_rel_gt']} "
          f"frac_floor={r['frac_floor']:.2e}; heat max_rel={rh['max_rel']:.3e} frac_floor={rh['frac_floor']:.2e}")
    assert r["n_sens"] == 0 and r["ok"] and r["support_equal"], r
    assert rh["ok"] and rh["support_equal"], rh
    m = cdh != 0
    assert np.array_equal(cdh[m], o["coldensh_out"][m])


@pytest.mark.parametrize("name", list(cases.CASES))
def test_cdh_last_source_vs_golden(asora, golden_dir, name, layout):
    """coldensh_out = outgoing column density of the last source: <= 1e-13 relative wherever the GPU
    visits (octahedron); the Fortran sweeps the whole cube, the GPU leaves zeros outside."""
    c = load_golden(golden_dir, name)
    phi, cdh, _ = run_hip(asora, c)
    ref = c["coldensh_last"]
    m = cdh != 0
    assert m.sum() > 0
    rel = np.abs(cdh[m] - ref[m]) / np.abs(ref[m])
    print(f"\n[{name}] cdh visited={m.sum()} max_rel={rel.max():.3e} bit-identical={np.array_equal(cdh[m], ref[m])}")
    assert rel.max() <= 1e-13
    N = c["N"]
    v, _ = asora.visit_count(c["R"], N)
    assert m.sum() == v, "number of visited cells differs from the analytic octahedron∩box count"


@pytest.mark.parametrize("name", ["uniform32_center", "lognormal48", "shadow32", "thick24"])
def test_heat_vs_golden(asora, golden_dir, name, layout):
    c = load_golden(golden_dir, name)
    assert bool(c["heat_valid"])
    phi, cdh, heat = run_hip(asora, c, heat=True)
    assert compare(phi, c["phi_ion"], sens=c["phi_sens"])["ok"]
    r = compare(heat, c["phi_heat"], sens=c["heat_sens"])
    print(f"\n[{name}] heat max_rel={r['max_rel']:.3e} frac_floor={r['frac_floor']:.2e} n_sens={r['n_sens']}")
    assert r["ok"], r


def test_reference_fixture_photo_ionization_rate(asora, port_oracle):
    """Inputs of the reference's own tests/test_asora.py::test_do_all_sources, checked against the
    oracle (the reference's .npy is a CUDA snapshot that differs from the Fortran on the
    Delta^2 == R^2 cells; it is not shipped to the GPU box)."""
    c = cases.reference_test_case()
    phi, cdh, _ = run_hip(asora, c)
    o = cases.run_oracle(port_oracle, c, heat=False)
    r = compare(phi, o["phi_ion"])
    print(f"\n[reference fixture] max_rel={r['max_rel']:.3e} frac_floor={r['frac_floor']:.2e}")
    assert r["ok"] and r["support_equal"], r
    assert phi.max() == pytest.approx(2.7e-39, rel=0.1)  # magnitude quoted in SURVEY.md section 4


@pytest.mark.parametrize("R", [0.0, 0.4, 1.0, 1.5, 2.0])
def test_tiny_radius_vs_oracle(asora, port_oracle, R):
    """R below one cell: only the source cell (and for R >= 1 its nearest neighbours) produce rates; the
    octahedron degenerates to q_max = 0, 1, 2, 3, 4 shells."""
    c = cases._base(12, R, 5e20, [[0, 11, 6], [3, 3, 3]], [1.0, 2.0], ndens=cases.lognormal_density(12, 4))
    phi, cdh, _ = run_hip(asora, c)
    o = cases.run_oracle(port_oracle, c, heat=False)
    r = compare(phi, o["phi_ion"])
    assert r["ok"] and r["support_equal"], r
    assert (phi != 0).sum() >= 2


@pytest.mark.parametrize("opts", [dict(), dict(quad_kernel=-1), dict(force_generic=1), dict(source_order=0),
                                  dict(R=4.0, quad_kernel=1, src_pack=2), dict(R=4.0, quad_kernel=2, src_pack=3)])
def test_source_positions_are_periodic(asora, port_oracle, ref_oracle, opts):
    """The reference is built with -DPERIODIC (src/asora/meson.build:33): source coordinates outside [0, N) -- the
    1-based lists of read_sources / generate_test_sourcefile contain N -- address cell (coordinate mod N), while the
    interpolation carries the coordinate as given (raytracing.f90:683).  Against the oracle fed the same list:
    the usual bar for the rates, bit-identical column density (which a wrap on upload would not give: the
    rounding of alam*di + real(i0) depends on i0)."""
    N = 24
    opts = dict(opts)
    R = opts.pop("R", 9.0)  # (R = 4: the multi-sector kernels with several sources per workgroup apply)
    c = cases._base(N, R, 5e20, [[N, -1, 5], [3, N + 2, -N - 3], [2 * N, 7, 7]], [1.0, 2.0, 0.5],
                    ndens=cases.lognormal_density(N, 4))
    for k, v in opts.items():
        asora.set_option(k, v)
    try:
        phi, cdh, _ = run_hip(asora, c)
        if "src_pack" in opts:
            assert asora.last_launch_info()["sources_per_workgroup"] == 1 << opts["src_pack"]
    finally:
        for k in opts:
            asora.set_option(k, {"source_order": 2}.get(k, 0))
    o = cases.run_reference(port_oracle, ref_oracle, c)
    r = compare(phi, o["phi_ion"], sens=o["phi_sens"])
    assert r["ok"] and r["support_equal"] and r["n_sens"] == 0, r
    m = cdh != 0
    assert m.sum() == asora.visit_count(c["R"], N)[0] and np.array_equal(cdh[m], o["coldensh_out"][m])
    # the wrapped list gives the same grid up to that rounding
    phi_w, _, _ = run_hip(asora, dict(c, pos=np.mod(c["pos"], N).astype(np.int32)))
    assert np.allclose(phi_w, phi, rtol=1e-11, atol=0) and np.array_equal(phi_w != 0, phi != 0)


def test_empty_source_list(asora, golden_dir):
    c = load_golden(golden_dir, "uniform32_center")
    phi, cdh, _ = run_hip(asora, c, num_src=0)
    assert not phi.any() and not cdh.any()


def test_cinterp_kat(golden_dir):
    """libasoratest.cinterp known answers (inputs of tests/test_asora_internal.py, outputs of the
    Fortran cinterp)."""
    from pyc2ray_amd import libasoratest
    z = np.load(os.path.join(golden_dir, "cinterp_kat.npz"))
    dens = np.random.default_rng(seed=42).random((20, 20, 20), dtype=np.float64)
    for pos0 in [(5, 5, 5), (1, 2, 3), (10, 5, 15)]:
        out = libasoratest.cinterp(pos0, dens)
        exp = z["%02d%02d%02d" % pos0]
        assert np.isnan(out[0, 0, 0]).all()
        m = ~np.isnan(exp)
        assert np.array_equal(out[m], exp[m]), np.max(np.abs(out[m] - exp[m]) / np.abs(exp[m]))
    out = libasoratest.cinterp_ex((0, 19, 7), dens, sigma=3.0, lo=-2, cnt=5)
    exp = z["wrap_001907_sigma3"]
    m = ~np.isnan(exp)
    assert np.array_equal(out[m], exp[m])


def test_exact_division_helpers(asora):
    """div_exact / rcp_exact / div_const are bit-identical to IEEE division on the ranges the path uses."""
    rng = np.random.default_rng(5)
    n = 1 << 20
    num = np.concatenate([rng.uniform(0, 1, n) * 10.0 ** rng.uniform(-120, 80, n),       # fluxes, rates, weights
                          rng.integers(0, 70000, n).astype(np.float64),                   # a^2 + b^2
                          rng.uniform(-5, 30, n)])                                       # logtau - minlogtau
    den = np.concatenate([rng.uniform(0.5, 2, n) * 10.0 ** rng.uniform(-30, 80, n),      # volumes, nHI, tau
                          rng.integers(1, 300, n).astype(np.float64) ** 2,                # n^2
                          np.full(n, 24.0 / 20000)])                                      # dlogtau
    assert asora.test_division(num, den) == (0, 0, 0)


@pytest.mark.parametrize("name", ["corner32", "lognormal48", "rcut7p3", "thick24"])
def test_table_driven_equals_table_free_kernel(asora, golden_dir, name):
    """The fast (geometry-table) kernel and the table-free kernel are the same arithmetic: for a single
    source (one writer per cell, no atomic-order effect) their outputs are bit-identical."""
    c = load_golden(golden_dir, name)
    phi_t, cdh_t, _ = run_hip(asora, c, num_src=1)
    assert asora.last_launch_info()["table_driven"]
    asora.set_option("force_generic", 1)
    try:
        phi_g, cdh_g, _ = run_hip(asora, c, num_src=1)
        assert not asora.last_launch_info()["table_driven"]
    finally:
        asora.set_option("force_generic", 0)
    assert np.array_equal(cdh_t, cdh_g)
    assert np.array_equal(phi_t, phi_g)


@pytest.mark.parametrize("opts", [dict(), dict(quad_kernel=-1), dict(quad_kernel=1), dict(quad_kernel=2, src_pack=2),
                                  dict(wg_threads=1024, compact_staging=1, quad_kernel=-1), dict(lds_stage_limit=1024)])
@pytest.mark.parametrize("name", ["corner33", "lognormal48", "rcut7p3", "same_cell_sources"])
def test_one_layout_equals_two_layouts(asora, golden_dir, name, opts):
    """Option "single_layout": the z-dominant sectors on the [i][j][k] grids (strided in-plane coordinate), rates added
    straight into the caller's grid.  Same arithmetic per cell, so a single source (one writer per cell) gives
    bit-identical rates, heating rates and column density in both modes, for every kernel shape."""
    c = load_golden(golden_dir, name)
    out = {}
    for k, v in opts.items():
        asora.set_option(k, v)
    try:
        for mode in (-1, 1):
            asora.set_option("single_layout", mode)
            out[mode] = run_hip(asora, c, heat=True, num_src=1)
            assert asora.last_launch_info()["single_layout"] == (mode == 1)
    finally:
        asora.set_option("single_layout", 0)
        for k in opts:
            asora.set_option(k, {"source_order": 2}.get(k, 0))
    for a, b in zip(out[-1], out[1]):
        assert np.array_equal(a, b)
    assert (out[1][0] != 0).sum() > 0 and (out[1][2] != 0).sum() > 0


def test_fallback_kernel_leaves_a_notice(asora, golden_dir):
    """The table-free kernel never runs silently: asora_last_launch_notice names the reason
    -- here the two that a small mesh can provoke, the option and the geometry-table budget; the 2^28-cell limit
    goes through the same code -- and is empty on the fast path."""
    c = load_golden(golden_dir, "rcut7p3")
    run_hip(asora, c, num_src=1)
    assert asora.last_launch_info()["notice"] == "" and asora.last_launch_info()["table_driven"]
    for opt, val, restore, word in (("force_generic", 1, 0, "force_generic"), ("max_table_cells", 10, 48 << 20, "max_table_cells")):
        asora.set_option(opt, val)
        try:
            run_hip(asora, c, num_src=1)
            info = asora.last_launch_info()
        finally:
            asora.set_option(opt, restore)
        assert not info["table_driven"] and word in info["notice"], info
    run_hip(asora, c, num_src=1)
    assert asora.last_launch_info()["notice"] == ""


@pytest.mark.parametrize("threads", [64, 128, 192, 512, 1024])
@pytest.mark.parametrize("name", ["corner32", "lognormal48", "rcut7p3", "thick24"])
def test_workgroup_sizes_equal_256_threads(asora, golden_dir, name, threads):
    """The library picks the workgroup size from the size of the staged entries (64 threads at small radii
    ... ONE 1024-thread workgroup per CU with the whole LDS at R ~ 100-125 cells); the variants differ only
    in how many 64-cell chunks make a phase, so a single source gives bit-identical output."""
    c = load_golden(golden_dir, name)
    asora.set_option("wg_threads", 256)
    try:
        phi_n, cdh_n, _ = run_hip(asora, c, num_src=1)
    finally:
        asora.set_option("wg_threads", 0)
    assert asora.last_launch_info()["block"] == 256
    asora.set_option("wg_threads", threads)
    try:
        phi_w, cdh_w, _ = run_hip(asora, c, num_src=1)
        info = asora.last_launch_info()
        assert info["block"] == threads and info["staging"] == "lds" and info["table_driven"], info
        phi_all, _, _ = run_hip(asora, c)
    finally:
        asora.set_option("wg_threads", 0)
    assert np.array_equal(cdh_w, cdh_n)
    assert np.array_equal(phi_w, phi_n)
    phi_ref, _, _ = run_hip(asora, c)  # automatic size, all sources
    assert np.allclose(phi_all, phi_ref, rtol=1e-13, atol=0)


def test_source_order_is_free(asora, golden_dir):
    """The library sweeps the list along a Hilbert curve (option source_order); phi is a sum over sources, so
    any order gives the same grid up to the rounding of the sum, and coldensh_out stays that of the caller's
    LAST source -- also when only a prefix of the uploaded list is raytraced."""
    c = load_golden(golden_dir, "multi16_unequal")
    out = {}
    for order in (0, 1, 2):
        asora.set_option("source_order", order)
        try:
            out[order] = run_hip(asora, c)
        finally:
            asora.set_option("source_order", 2)
    for order in (1, 2):
        assert np.array_equal(out[order][1], out[0][1])                       # cdh of the last source
        assert np.allclose(out[order][0], out[0][0], rtol=1e-13, atol=0)
    # prefix of the uploaded list: same as uploading only the prefix
    N, ns, k = c["N"], len(c["flux"]), 5
    asora.device_init(N, 8, 0, 1)
    try:
        asora.photo_table_to_device(np.ascontiguousarray(c["thin"]), np.ascontiguousarray(c["thick"]), int(c["num_tau"]))
        asora.density_to_device(np.ravel(c["ndens"]).astype(np.float64), N)
        pos = np.ascontiguousarray(np.asarray(c["pos"], dtype=np.int32).reshape(-1, 3).ravel())
        asora.source_data_to_device(pos, np.ascontiguousarray(c["flux"], dtype=np.float64), ns)
        xh = np.ravel(c["xh"]).astype(np.float64)
        res = []
        for n in (k, ns, k):
            cdh, phi = np.zeros(N ** 3), np.zeros(N ** 3)
            asora.do_all_sources(c["R"], cdh, c["sig"], c["dr"], None, xh, phi, n, N, c["minlogtau"], c["dlogtau"],
                                 int(c["num_tau"]))
            res.append((phi, cdh))
    finally:
        asora.device_close()
    phi_k, cdh_k, _ = run_hip(asora, c, num_src=k)
    for i in (0, 2):
        assert np.array_equal(res[i][1].reshape(N, N, N), cdh_k)
        assert np.allclose(res[i][0].reshape(N, N, N), phi_k, rtol=1e-13, atol=0)
    assert np.array_equal(res[1][1].reshape(N, N, N), out[2][1])


@pytest.mark.parametrize("threads", [256, 512, 1024])
@pytest.mark.parametrize("name", ["corner33", "lognormal48", "rcut7p3", "thick24"])
def test_compact_staging_equals_pair_staging(asora, golden_dir, name, threads):
    """Column-density-only staging (picked when the (column density, weight) pairs do not fit in LDS) recomputes
    the weight 1/max(0.6, tau) from the same staged value with the same operation: bit-identical output."""
    c = load_golden(golden_dir, name)
    phi_n, cdh_n, _ = run_hip(asora, c, num_src=1)
    assert asora.last_launch_info()["staging"] == "lds"
    asora.set_option("compact_staging", 1)
    asora.set_option("wg_threads", threads)
    try:
        phi_w, cdh_w, _ = run_hip(asora, c, num_src=1)
        info = asora.last_launch_info()
        assert info["block"] == threads and info["staging"] == "lds-compact" and info["table_driven"], info
    finally:
        asora.set_option("compact_staging", 0)
        asora.set_option("wg_threads", 0)
    assert np.array_equal(cdh_w, cdh_n)
    assert np.array_equal(phi_w, phi_n)


@pytest.mark.parametrize("opts", [dict(wg_threads=64), dict(wg_threads=1024), dict(wg_threads=512, compact_staging=1),
                                  dict(skip_unobserved=1), dict(source_order=0)])
def test_heat_output_of_the_kernel_variants(asora, golden_dir, opts):
    """The heating rate rides on the same kernel variants: single source -> bit-identical phi, heat and
    coldensh_out whatever workgroup size / staging layout / sweep options are picked."""
    c = load_golden(golden_dir, "lognormal48")
    assert bool(c["heat_valid"])
    ref = run_hip(asora, c, heat=True, num_src=1)
    for k, v in opts.items():
        asora.set_option(k, v)
    try:
        out = run_hip(asora, c, heat=True, num_src=1)
    finally:
        for k in opts:
            asora.set_option(k, 2 if k == "source_order" else 0)
    for a, b in zip(out, ref):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("opts", [dict(wg_threads=64), dict(wg_threads=128), dict(wg_threads=256),
                                  dict(wg_threads=128, compact_staging=1), dict(),
                                  dict(quad_kernel=2), dict(quad_kernel=2, wg_threads=64),
                                  dict(quad_kernel=2, wg_threads=256, compact_staging=1)])
@pytest.mark.parametrize("name", ["rcut7p3", "multi16_equal", "multi16_unequal"])
def test_four_quadrant_kernel_equals_sector_kernel(asora, golden_dir, name, opts):
    """The four-quadrant and eight-octant variants (one workgroup per source and pyramid / axis, lane l on sector
    l & 3 / l & 7; picked for small radii) do the same arithmetic per cell: single source -> bit-identical phi, heat and coldensh_out; all
    sources -> the same sum."""
    c = load_golden(golden_dir, name)
    heat = bool(c["heat_valid"])
    asora.set_option("quad_kernel", -1)
    try:
        ref = run_hip(asora, c, heat=heat, num_src=1)
        assert not asora.last_launch_info()["quad"]
        ref_all = run_hip(asora, c)
    finally:
        asora.set_option("quad_kernel", 0)
    asora.set_option("quad_kernel", 1)
    for k, v in opts.items():
        asora.set_option(k, v)
    try:
        out = run_hip(asora, c, heat=heat, num_src=1)
        info = asora.last_launch_info()
        assert info["quad"] and info["table_driven"] and info["staging"].startswith("lds"), info
        if name == "rcut7p3":  # (R = 10 at N = 32 reaches the box limits: four quadrants, not eight octants)
            assert info["sectors_per_workgroup"] == (8 if opts.get("quad_kernel") == 2 else 4), info
        out_all = run_hip(asora, c)
    finally:
        asora.set_option("quad_kernel", 0)
        for k in opts:
            asora.set_option(k, 0)
    for a, b in zip(out, ref):
        assert (a is None and b is None) or np.array_equal(a, b)
    assert np.array_equal(out_all[1], ref_all[1])
    assert np.allclose(out_all[0], ref_all[0], rtol=1e-13, atol=0)


@pytest.mark.parametrize("opts", [dict(quad_kernel=1, src_pack=2), dict(quad_kernel=1, src_pack=3),
                                  dict(quad_kernel=2, src_pack=2), dict(quad_kernel=2, src_pack=3),
                                  dict(quad_kernel=1, src_pack=2, skip_unobserved=1), dict(quad_kernel=2, src_pack=3, source_order=0)])
@pytest.mark.parametrize("name", ["rcut7p3", "multi16_equal", "multi16_unequal", "same_cell_sources"])
def test_source_packing_equals_sector_kernel(asora, golden_dir, name, opts):
    """Several sources per workgroup (option src_pack: lane l on quadrant l & 3 / octant l & 7 of source (l >> 2 / 3) &
    3 / 7 of its group; picked for the smallest radii) do the same arithmetic per cell and per source: single source and
    every list length -- full groups, a partial last group -- give bit-identical column density and the same rate sum."""
    c = load_golden(golden_dir, name)
    heat = bool(c["heat_valid"])
    ns_all = len(c["flux"])
    asora.set_option("quad_kernel", -1)
    try:
        ref = {n: run_hip(asora, c, heat=heat, num_src=n) for n in sorted({1, min(5, ns_all), ns_all})}
        assert not asora.last_launch_info()["quad"]
    finally:
        asora.set_option("quad_kernel", 0)
    for k, v in opts.items():
        asora.set_option(k, v)
    try:
        for n, r in ref.items():
            out = run_hip(asora, c, heat=heat, num_src=n)
            info = asora.last_launch_info()
            assert info["sources_per_workgroup"] == 1 << opts["src_pack"] and info["block"] == 256, info
            if name == "rcut7p3":
                assert info["sectors_per_workgroup"] == (8 if opts["quad_kernel"] == 2 else 4), info
            assert info["grid"] == (24 // info["sectors_per_workgroup"]) * -(-n // info["sources_per_workgroup"]), info
            assert np.array_equal(out[1], r[1])                                   # column density of the last source
            for a, b in ((out[0], r[0]), (out[2], r[2])):
                if a is None:
                    continue
                assert np.array_equal(a != 0, b != 0)
                if n == 1:
                    assert np.array_equal(a, b)
                else:
                    assert np.allclose(a, b, rtol=1e-13, atol=0)
    finally:
        for k in opts:
            asora.set_option(k, {"source_order": 2}.get(k, 0))


@pytest.mark.parametrize("opts", [dict(quad_kernel=1), dict(quad_kernel=2), dict(quad_kernel=1, wg_threads=256, compact_staging=1),
                                  dict(quad_kernel=2, src_pack=2)])
@pytest.mark.parametrize("name", ["multi16_unequal", "rcut7p3"])
def test_staging_array_stride_is_free(asora, golden_dir, name, opts):
    """The multi-sector kernels keep one staging array per sector (and source), E entries apart; the host pads E so that
    the lanes of a cell hit different LDS banks (option stride_residue: -1 automatic, -2 none, >= 0 forced modulo 32).
    The padding entries are never addressed: every choice gives the same bits."""
    c = load_golden(golden_dir, name)
    for k, v in opts.items():
        asora.set_option(k, v)
    out, lds = {}, {}
    try:
        for sr in (-2, -1, 0, 5, 31):
            asora.set_option("stride_residue", sr)
            out[sr] = (run_hip(asora, c, heat=bool(c["heat_valid"]), num_src=1), run_hip(asora, c))
            info = asora.last_launch_info()
            assert info["quad"], info
            lds[sr] = info["lds_bytes"]
    finally:
        asora.set_option("stride_residue", -1)
        for k in opts:
            asora.set_option(k, 0)
    print(name, opts, "LDS bytes by stride_residue:", lds)  # (small arrays: a pad that changes the workgroups per CU is not applied)
    one, full = out[-2]
    for sr, (o, f) in out.items():
        for a, b in zip(o, one):
            assert (a is None and b is None) or np.array_equal(a, b), sr
        assert np.array_equal(f[1], full[1]), sr
        assert np.allclose(f[0], full[0], rtol=1e-13, atol=0), sr


@pytest.mark.parametrize("name", ["multi16_unequal", "rcut7p3", "lognormal48", "corner33"])
def test_skip_unobserved_cells(asora, golden_dir, name):
    """Option skip_unobserved (off by default; never used by bench.py): cells outside the R-sphere are not
    upwind of any cell inside it, so they only matter for the exported column density (the last source's).
    With the option every other source sweeps the rate-producing cells only: same phi, same coldensh_out."""
    c = load_golden(golden_dir, name)
    phi0, cdh0, _ = run_hip(asora, c)
    asora.set_option("skip_unobserved", 1)
    try:
        phi1, cdh1, _ = run_hip(asora, c)
    finally:
        asora.set_option("skip_unobserved", 0)
    assert np.array_equal(cdh1, cdh0)
    assert np.array_equal(phi1 != 0, phi0 != 0)
    assert np.allclose(phi1, phi0, rtol=1e-13, atol=0)


@pytest.mark.parametrize("opts", [dict(quad_kernel=-1), dict(quad_kernel=-1, wg_threads=64), dict(quad_kernel=-1, wg_threads=1024, compact_staging=1),
                                  dict(quad_kernel=-1, skip_unobserved=1), dict(lds_stage_limit=1024), dict()])
@pytest.mark.parametrize("name", ["thick24", "capped16", "shadow32", "lognormal48", "multi16_unequal"])
def test_skip_saturated_cells(asora, golden_dir, name, opts):
    """Option skip_saturated (off by default; never used by bench.py): behind an optical depth beyond the table's last
    entry both look-ups of a thick cell clamp to that entry (photorates.f90:140-146), its rate is exactly +0, and the
    kernel skips the evaluation.  Same numbers: single source bit-identical (phi, heat, coldensh_out), all sources the
    same sum and support; thick24 (tau ~ 290 per cell) is saturated from the 35th cell on."""
    c = load_golden(golden_dir, name)
    heat = bool(c["heat_valid"])
    for k, v in opts.items():
        asora.set_option(k, v)
    try:
        ref1, ref = run_hip(asora, c, heat=heat, num_src=1), run_hip(asora, c, heat=heat)
        asora.set_option("skip_saturated", 1)
        out1, out = run_hip(asora, c, heat=heat, num_src=1), run_hip(asora, c, heat=heat)
    finally:
        asora.set_option("skip_saturated", 0)
        for k in opts:
            asora.set_option(k, 0)
    for a, b in zip(out1, ref1):
        assert (a is None and b is None) or np.array_equal(a, b)
    assert np.array_equal(out[1], ref[1])
    for a, b in ((out[0], ref[0]), (out[2], ref[2])):
        if a is not None:
            assert np.array_equal(a != 0, b != 0) and np.allclose(a, b, rtol=1e-13, atol=0)


def test_error_reporting(asora):
    """Errors surface as RuntimeError instead of being swallowed."""
    with pytest.raises(RuntimeError):
        asora.do_all_sources(1.0, np.zeros(8), 1.0, 1.0, None, np.zeros(8), np.zeros(8), 1, 2, -20.0, 0.1, 10)
    asora.device_init(8, 1, 0, 1)
    try:
        with pytest.raises(RuntimeError):  # tables missing
            asora.do_all_sources(1.0, np.zeros(512), 1.0, 1.0, None, np.zeros(512), np.zeros(512), 0, 8, -20.0, 0.1, 10)
        asora.source_data_to_device(np.array([0, 0, 8], dtype=np.int32), np.ones(1), 1)  # periodic, like the reference
        with pytest.raises(RuntimeError):  # not a grid position at all
            asora.source_data_to_device(np.array([0, 0, 1 << 30], dtype=np.int32), np.ones(1), 1)
        with pytest.raises(TypeError):
            asora.density_to_device(np.zeros(512, dtype=np.float32), 8)
    finally:
        asora.device_close()
