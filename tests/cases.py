"""Seeded input generators shared by the golden-vector script, the CPU tests and the GPU parity tests.

Every case is a dict of plain numpy inputs for one ``do_all_sources`` call:
  N, R, dr, sig, minlogtau, dlogtau, num_tau, thin, thick, heat_thin, heat_thick,
  ndens (N,N,N), xh (N,N,N), pos (Ns,3) 0-based int32, flux (Ns,)
The list follows SURVEY.md 8(c) "golden vectors to generate" (1)-(6) plus a capped and an
optically very thick case.  Nothing here reads /root/reference.
"""
from __future__ import annotations

import functools

import numpy as np

from pyc2ray_amd.radiation import EV, H_PLANCK, BlackBodySource, make_tau_table

PC = 3.0856775814913673e18
MPC = 3.0856775814913673e24
SIGMA_HI = 6.30e-18
FREQ_MIN = 13.598 * EV / H_PLANCK
FREQ_MAX = 54.416 * EV / H_PLANCK


@functools.lru_cache(maxsize=None)
def tables(temp=5e4, pl_index=2.8, minlogtau=-20.0, maxlogtau=4.0, numtau=2000):
    """(thin, thick, heat_thin, heat_thick, minlogtau, dlogtau); tables have numtau+1 entries."""
    tau, dlogtau = make_tau_table(minlogtau, maxlogtau, numtau)
    bb = BlackBodySource(temp, False, FREQ_MIN, pl_index)
    thin, thick = bb.make_photo_table(tau, FREQ_MIN, FREQ_MAX, 1e48)
    bb = BlackBodySource(temp, False, FREQ_MIN, pl_index)
    hthin, hthick = bb.make_heat_table(tau, FREQ_MIN, FREQ_MAX, 1e48)
    return thin, thick, hthin, hthick, minlogtau, dlogtau


def lognormal_density(N, seed, mean=1e-3, sigma_ln=1.0, smooth=1.5):
    """Log-normal field with a little spatial correlation (periodic Gaussian smoothing in k-space)."""
    rng = np.random.default_rng(seed)
    g = rng.standard_normal((N, N, N))
    k = np.fft.fftfreq(N)
    k2 = k[:, None, None] ** 2 + k[None, :, None] ** 2 + k[None, None, :] ** 2
    g = np.fft.ifftn(np.fft.fftn(g) * np.exp(-2.0 * (np.pi * smooth) ** 2 * k2)).real
    g = (g - g.mean()) / g.std()
    d = np.exp(sigma_ln * g)
    return np.ascontiguousarray(mean * d / d.mean())


def _base(N, R, dr, pos, flux, ndens=None, xh=None, table_kw=None, num_tau_arg="shape"):
    thin, thick, hthin, hthick, minlogtau, dlogtau = tables(**(table_kw or {}))
    c = dict(N=N, R=float(R), dr=float(dr), sig=SIGMA_HI, minlogtau=minlogtau, dlogtau=dlogtau,
             thin=thin, thick=thick, heat_thin=hthin, heat_thick=hthick,
             # the reference's callers pass either table.shape[0] (asora_core.py:59) or NumTau
             # (tests/test_asora.py:57); both conventions are exercised
             num_tau=thin.shape[0] if num_tau_arg == "shape" else thin.shape[0] - 1,
             ndens=np.full((N, N, N), 1e-3) if ndens is None else np.ascontiguousarray(ndens, dtype=np.float64),
             xh=np.full((N, N, N), 1e-4) if xh is None else np.ascontiguousarray(xh, dtype=np.float64),
             pos=np.asarray(pos, dtype=np.int32).reshape(-1, 3),
             flux=np.asarray(flux, dtype=np.float64).reshape(-1))
    return c


def case_uniform32_center():
    # (1) uniform, centre source, whole box; cell = 14 kpc / 128 as in config 1 of BASELINE.json
    return _base(32, 1e3, 0.014 * MPC / 128, [[16, 16, 16]], [10.0])


def case_corner32():
    # (2) source next to a corner: periodic wrap, even N (asymmetric last_l/last_r)
    return _base(32, 1e3, 0.014 * MPC / 128, [[1, 30, 0]], [3.0], num_tau_arg="numtau")


def case_corner33():
    # (2) odd N: last_r = N/2 - 1 + N%2
    return _base(33, 1e3, 0.014 * MPC / 128, [[32, 0, 31]], [3.0])


def case_lognormal48():
    # (3) log-normal density, random ionised fraction incl. almost fully ionised cells:
    # exercises the thin branch and the thick-branch cancellation
    N = 48
    rng = np.random.default_rng(1234)
    xh = rng.uniform(1e-4, 0.9999, (N, N, N))
    xh[rng.random((N, N, N)) < 0.3] = 1.0 - 1e-7
    nd = lognormal_density(N, 7)
    pos = rng.integers(0, N, (3, 3))
    return _base(N, 1e3, 0.02 * MPC / N, pos, [5.0, 1.0, 20.0], ndens=nd, xh=xh)


def case_shadow32():
    # (4) dense clump casting a shadow (cf. examples/paper_tests/test4_shadow/shadow.py:34-45)
    N = 32
    nd = np.full((N, N, N), 1e-3)
    ii, jj, kk = np.meshgrid(np.arange(N), np.arange(N), np.arange(N), indexing="ij")
    nd[(ii - 20) ** 2 + (jj - 20) ** 2 + (kk - 15) ** 2 <= 9] *= 600.0
    return _base(N, 1e3, 0.05 * MPC / N, [[14, 14, 15]], [10.0], ndens=nd)


def case_rcut7p3():
    # (5) non-integer R smaller than the box: dist2 > R^2 cut and q_max rounding
    return _base(32, 7.3, 0.014 * MPC / 128, [[5, 27, 16], [16, 16, 16]], [2.0, 7.0])


def case_multi16_equal():
    # (6) 16 sources, equal flux, integer R (cells exactly on the R sphere: SURVEY 3.4 #11)
    rng = np.random.default_rng(918)
    pos = rng.integers(0, 32, (16, 3))
    return _base(32, 10.0, 50 * PC / 50, pos, np.full(16, 2.5))


def case_multi16_unequal():
    rng = np.random.default_rng(919)
    pos = rng.integers(0, 32, (16, 3))
    flux = rng.uniform(1e10, 1e14, 16) * 100 / 1e48
    nd = lognormal_density(32, 11)
    return _base(32, 10.0, 0.3 * MPC / 32, pos, flux, ndens=nd, num_tau_arg="numtau")


def case_thick24():
    # optically very thick cells (benchmark cell size 3 Mpc/200, tau ~ 290 per cell):
    # runs off the end of the table
    return _base(24, 1e3, 3.0 * MPC / 200, [[3, 12, 20]], [1.0e3])


def case_capped16():
    # column density beyond max_coldensh = 2e30 (raytracing.f90:368,478): rates stop, cdh goes on
    N = 16
    nd = np.full((N, N, N), 1e-3)
    nd[6:9, 6:9, 6:9] = 1e9
    return _base(N, 1e3, 3.0 * MPC / 200, [[7, 7, 7]], [1.0])


def case_same_cell_sources():
    # several sources in the same cell and in neighbouring cells (halo lists do this)
    pos = [[4, 4, 4], [4, 4, 4], [5, 4, 4], [4, 4, 4]]
    return _base(24, 6.0, 0.1 * MPC / 24, pos, [1.0, 2.0, 3.0, 4.0])


CASES = {
    "uniform32_center": case_uniform32_center,
    "corner32": case_corner32,
    "corner33": case_corner33,
    "lognormal48": case_lognormal48,
    "shadow32": case_shadow32,
    "rcut7p3": case_rcut7p3,
    "multi16_equal": case_multi16_equal,
    "multi16_unequal": case_multi16_unequal,
    "thick24": case_thick24,
    "capped16": case_capped16,
    "same_cell_sources": case_same_cell_sources,
}


def reference_test_case(N=50, ns=10, R=15.0, numtau=20000):
    """The inputs of the reference's own integration test (tests/test_asora.py:33-97)."""
    tau, dlogtau = make_tau_table(-20.0, 4.0, numtau)
    # the reference passes sigma as the power-law index (test_asora.py:51) -> grey to 1e-17
    bb = BlackBodySource(1e5, False, FREQ_MIN, SIGMA_HI)
    thin, thick = bb.make_photo_table(tau, FREQ_MIN, FREQ_MAX, 1e48)
    rng = np.random.default_rng(918)
    src_pos = rng.integers(0, N, size=(3 * ns), dtype=np.int32)
    flux = rng.uniform(1e10, 1e14, size=ns).astype(np.float64) * 100.0 / 1e48
    return dict(N=N, R=R, dr=50 * PC / N, sig=SIGMA_HI, minlogtau=-20.0, dlogtau=dlogtau, num_tau=numtau,
                thin=thin, thick=thick, heat_thin=np.zeros_like(thin), heat_thick=np.zeros_like(thick),
                ndens=np.full((N, N, N), 1e-3), xh=np.full((N, N, N), 1e-4),
                pos=src_pos.reshape(ns, 3), flux=flux, pos_flat=src_pos)


YEAR = 3.15576e7
# chemistry constants of the reference's parameter files (examples/*/parameters.yml CGS / Abundances
# sections, combined as in pyc2ray/c2ray_base.py:547-556)
CHEM = dict(bh00=2.59e-13, albpow=-0.7, colh0=1.3e-8 * 0.83 * 1.0 / 13.598 ** 2, temph0=13.598 * 11604.51812,
            abu_c=7.1e-7)


def evolve_case(N=32, seed=5):
    """One evolve step (SURVEY.md 8(c) golden (8)): three unequal sources switching on in an almost
    neutral log-normal medium, dt = 1 Myr, cell of the single-source hackathon test (0.014 Mpc / 128),
    T = 1e4 K, clumping 1, convergence_fraction 1e-4.  The ionisation fronts travel ~10 cells, so the
    loop needs several raytrace/chemistry iterations."""
    rng = np.random.default_rng(seed)
    c = _base(N, 1.5 * N, 0.014 * MPC / 128, [[N // 2, N // 2, N // 2], [5, 27, 9], [22, 6, 25]], [10.0, 2.5, 0.4],
              ndens=lognormal_density(N, seed, sigma_ln=0.5), xh=np.full((N, N, N), 1.2e-3))
    c.update(CHEM)
    c.update(dt=1e6 * YEAR, temp=np.full((N, N, N), 1e4), clump=np.ones((N, N, N)), convergence_fraction=1e-4)
    return c


def evolve_case_shadow128():
    """BASELINE config 3 (examples/paper_tests/test4_shadow): 128^3, cell 0.014 Mpc / 128,
    one 1e49 s^-1 source at (64, 64, 64), ndens 1e-3 with a x6 over-dense sphere of radius 8 cells centred
    (76, 76, 63) (0-based), xh0 = 1.2e-3, T = 1e4 K, black body 5e4 K, R_max = 150 cells (full box),
    time steps of 1 Myr."""
    N = 128
    nd = np.full((N, N, N), 1e-3)
    i, j, k = np.meshgrid(np.arange(N), np.arange(N), np.arange(N), indexing="ij")
    nd[(i - 76) ** 2 + (j - 76) ** 2 + (k - 63) ** 2 <= 64] = 6e-3
    c = _base(N, 150.0, 0.014 * MPC / 128, [[64, 64, 64]], [10.0], ndens=nd, xh=np.full((N, N, N), 1.2e-3),
              table_kw=dict(numtau=10000))
    c.update(CHEM)
    c.update(dt=1e6 * YEAR, temp=np.full((N, N, N), 1e4), clump=np.ones((N, N, N)), convergence_fraction=1e-4)
    return c


def chemistry_case(N=24, seed=3):
    """Random per-cell inputs for one chemistry pass covering neutral .. fully ionised cells, zero and
    huge rates, temperatures 1e2 .. 1e5 K, clumping 1 .. 30, and a time average that differs from xh."""
    rng = np.random.default_rng(seed)
    sh = (N, N, N)
    xh = 10.0 ** rng.uniform(-4, 0, sh)
    xh[rng.random(sh) < 0.05] = 1.0 - 1e-9
    xh_av = np.clip(xh * 10.0 ** rng.uniform(-0.3, 0.3, sh), 1e-14, 1.0 - 1e-12)
    phi = 10.0 ** rng.uniform(-18, -9, sh)
    phi[rng.random(sh) < 0.2] = 0.0
    c = dict(N=N, dt=10.0 ** rng.uniform(12, 14.5), ndens=10.0 ** rng.uniform(-5, -1, sh),
             temp=10.0 ** rng.uniform(2, 5, sh), xh=xh, xh_av=xh_av, xh_intermed=xh.copy(), phi_ion=phi,
             clump=rng.uniform(1.0, 30.0, sh))
    c.update(CHEM)
    return c


def run_oracle(oracle, c, heat=True, flux_mode=1, **kw):
    return oracle.do_all_sources(c["flux"], c["pos"], c["ndens"], c["xh"], c["sig"], c["dr"], c["thin"],
                                 c["thick"], c["minlogtau"], c["dlogtau"], c["R"],
                                 heat_thin=c["heat_thin"] if heat else None,
                                 heat_thick=c["heat_thick"] if heat else None, num_tau=c["num_tau"],
                                 flux_mode=flux_mode, **kw)


def run_reference(port, ref, c, heat=False, **kw):
    """What the full-size GPU tests compare against: the COMPILED REFERENCE's rates when oracle/_ref is there
    (`ref` not None), else the port's; the conditioning term `phi_sens` always comes from the port (the
    reference has no such output).  With both at hand the port is first held to the reference on this very
    input: same support, rates <= 1e-12 relative (libm log10 variants only), column density bit-identical."""
    o = run_oracle(port, c, heat=heat, sens=True, **kw)
    o["kind"] = "port"
    if ref is None:
        return o
    r = run_oracle(ref, c, heat=heat, **kw)
    for k in ("phi_ion",) + (("phi_heat",) if heat else ()):
        m = r[k] != 0
        assert np.array_equal(o[k] != 0, m), f"port and compiled reference differ in the support of {k}"
        assert np.max(np.abs(o[k][m] - r[k][m]) / np.abs(r[k][m])) <= 1e-12
        o[k] = r[k]
    assert np.array_equal(o["coldensh_out"], r["coldensh_out"])
    o["kind"] = "reference"
    return o
