"""Generate the committed golden vectors from the REFERENCE Fortran (build container only).

    python tests/golden/make_golden.py

needs oracle/_ref/libc2ray_ref.so (``make -C oracle ref``: flang build of the reference's
src/c2ray/*.f90, compiled where they lie under /root/reference).  For every case in
tests/cases.py it runs the reference one source per call and stores inputs +
outputs in tests/golden/<case>.npz; it also stores the cinterp known-answer vectors (same inputs
as the reference's tests/test_asora_internal.py:10-20, outputs from the Fortran ``cinterp``) and
one full evolve step (raytrace + chemistry.global_pass iterations).

It refuses to write a file when our C restatement (oracle/c2ray_oracle.c) disagrees with the
reference: cdh must be bit-identical, phi within 1e-13 relative (same libm here, so in practice
also bit-identical -- the margin only covers libm ifunc variants between machines).
"""
from __future__ import annotations

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import cases  # noqa: E402
from oracle.oracle import PortOracle, RefOracle, build_ref  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
HEAT_VALID = {"uniform32_center", "corner32", "corner33", "lognormal48", "shadow32", "thick24"}


def relerr(a, b):
    m = np.abs(b) > 0
    e = np.zeros_like(a)
    e[m] = np.abs(a[m] - b[m]) / np.abs(b[m])
    e[~m] = np.abs(a[~m])
    return e.max() if e.size else 0.0


def main():
    build_ref()
    ref, port = RefOracle(), PortOracle()
    for name, fn in cases.CASES.items():
        c = fn()
        r = cases.run_oracle(ref, c)
        p = cases.run_oracle(port, c, sens=True)
        assert np.array_equal(r["coldensh_out"], p["coldensh_out"]), name
        e = relerr(p["phi_ion"], r["phi_ion"])
        assert e <= 1e-13, (name, e)
        if name in HEAT_VALID:
            assert relerr(p["phi_heat"], r["phi_heat"]) <= 1e-13, name
        # the reference bug mode (every source uses the LAST flux) is pinned too
        r0 = cases.run_oracle(ref, c, flux_mode=0)
        p0 = cases.run_oracle(port, c, flux_mode=0)
        assert relerr(p0["phi_ion"], r0["phi_ion"]) <= 1e-13, name
        np.savez_compressed(
            os.path.join(OUT, name + ".npz"),
            N=c["N"], R=c["R"], dr=c["dr"], sig=c["sig"], minlogtau=c["minlogtau"], dlogtau=c["dlogtau"],
            num_tau=c["num_tau"], thin=c["thin"], thick=c["thick"], heat_thin=c["heat_thin"],
            heat_thick=c["heat_thick"], ndens=c["ndens"], xh=c["xh"], pos=c["pos"], flux=c["flux"],
            phi_ion=r["phi_ion"], phi_heat=r["phi_heat"], heat_valid=name in HEAT_VALID,
            coldensh_last=r["coldensh_out"], phi_ion_lastflux_bug=r0["phi_ion"],
            photon_loss=r["photon_loss"], nsubbox=r["nsubbox"],
            # conditioning of the reference values (from the port, see oracle/c2ray_oracle.c
            # c2ray_photoion_rates_sens); float32 is plenty for a tolerance term
            phi_sens=p["phi_sens"].astype(np.float32), heat_sens=p["heat_sens"].astype(np.float32))
        print(f"{name:22s} N={c['N']:3d} Ns={len(c['flux']):2d} port-vs-ref phi rel {e:.1e}  max phi {r['phi_ion'].max():.3e}")

    # cinterp known-answer vectors: inputs of the reference test, outputs of the Fortran cinterp
    rng = np.random.default_rng(seed=42)
    dens = rng.random((20, 20, 20), dtype=np.float64)
    kat = {}
    for pos0 in [(5, 5, 5), (1, 2, 3), (10, 5, 15)]:
        out = np.full((4, 4, 4, 2), np.nan)
        for i in range(4):
            for j in range(4):
                for k in range(4):
                    if i == j == k == 0:
                        continue
                    out[i, j, k] = ref.cinterp((pos0[0] + i, pos0[1] + j, pos0[2] + k), pos0, dens, 1.0)
                    assert tuple(out[i, j, k]) == port.cinterp((pos0[0] + i, pos0[1] + j, pos0[2] + k), pos0, dens, 1.0)
        kat["%02d%02d%02d" % pos0] = out
    # negative offsets / wrap / other sigma, not covered by the reference's test
    pos0 = (0, 19, 7)
    out = np.full((5, 5, 5, 2), np.nan)
    for i in range(-2, 3):
        for j in range(-2, 3):
            for k in range(-2, 3):
                if i == j == k == 0:
                    continue
                out[i + 2, j + 2, k + 2] = ref.cinterp((pos0[0] + i, pos0[1] + j, pos0[2] + k), pos0, dens, 3.0)
    kat["wrap_001907_sigma3"] = out
    np.savez_compressed(os.path.join(OUT, "cinterp_kat.npz"), **kat)
    print("cinterp_kat.npz written")

    make_evolve_golden(ref, port)


def make_evolve_golden(ref, port):
    """(8) one full evolve step at 32^3 and one stand-alone chemistry pass, from the reference."""
    from evolve_host import evolve3D_host  # host loop of the product, oracle injected

    c = cases.evolve_case()
    a, b = evolve3D_host(ref, c), evolve3D_host(port, c)
    assert a["niter"] == b["niter"] and a["conv_flags"] == b["conv_flags"]
    assert np.max(np.abs(a["xh_new"] - b["xh_new"])) <= 1e-14
    np.savez_compressed(os.path.join(OUT, "evolve32.npz"), xh_new=a["xh_new"], xh_av=a["xh_av"], phi_ion=a["phi_ion"],
                        niter=a["niter"], conv_flags=np.asarray(a["conv_flags"]), phi_first=a["phi_first"])
    print("evolve32.npz written: niter", a["niter"], "conv_flags", a["conv_flags"],
          "mean xh", float(a["xh_new"].mean()))

    c = cases.chemistry_case()
    out = {}
    for tag, orc in (("ref", ref), ("port", port)):
        xa, xi = c["xh_av"].copy(), c["xh_intermed"].copy()
        cf = orc.global_pass(c["dt"], c["ndens"], c["temp"], c["xh"], xa, xi, c["phi_ion"], c["clump"], c["bh00"],
                             c["albpow"], c["colh0"], c["temph0"], c["abu_c"])
        out[tag] = (cf, xa, xi)
    assert out["ref"][0] == out["port"][0]
    assert np.array_equal(out["ref"][1], out["port"][1]) and np.array_equal(out["ref"][2], out["port"][2])
    np.savez_compressed(os.path.join(OUT, "chemistry24.npz"), conv_flag=out["ref"][0], xh_av=out["ref"][1],
                        xh_intermed=out["ref"][2])
    print("chemistry24.npz written: conv_flag", out["ref"][0])


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "evolve":
        build_ref()
        make_evolve_golden(RefOracle(), PortOracle())
    else:
        main()
