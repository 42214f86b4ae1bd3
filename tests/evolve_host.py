"""The evolve loop of pyc2ray_amd.evolve driven by the CPU oracle -- TEST INFRASTRUCTURE.

Same host logic as the product (``pyc2ray_amd.evolve._evolve_loop``: iteration, convergence
criterion, relative-change tests), with the two device operations replaced by the oracle's
raytrace (one source per call, per-source flux, full box: the ASORA semantics)
and the oracle's ``global_pass``.  Used to generate tests/golden/evolve32.npz from the compiled
reference and to check the C restatement against it.
"""
import numpy as np

from pyc2ray_amd.evolve import _evolve_loop


def evolve3D_host(oracle, c, history=None):
    N = c["N"]
    ndens, temp, xh, clump = (np.ascontiguousarray(c[k], dtype=np.float64) for k in ("ndens", "temp", "xh", "clump"))
    xh_av = xh.copy()
    xh_intermed = xh.copy()
    state = dict(phi=None, phis=[])

    def raytrace():
        o = oracle.do_all_sources(c["flux"], c["pos"], ndens, xh_av, c["sig"], c["dr"], c["thin"], c["thick"],
                                  c["minlogtau"], c["dlogtau"], c["R"], num_tau=c["num_tau"], flux_mode=1)
        state["phi"] = np.ascontiguousarray(o["phi_ion"])
        state["phis"].append(state["phi"].copy())

    def chemistry():
        conv_flag = oracle.global_pass(c["dt"], ndens, temp, xh, xh_av, xh_intermed, state["phi"], clump,
                                       c["bh00"], c["albpow"], c["colh0"], c["temph0"], c["abu_c"])
        return conv_flag, np.sum(xh_intermed), np.sum(1.0 - xh_intermed)

    hist = [] if history is None else history
    niter = _evolve_loop(raytrace, chemistry, N ** 3, len(c["flux"]), c["convergence_fraction"], history=hist)
    return dict(xh_new=xh_intermed, xh_av=xh_av, phi_ion=state["phi"], niter=niter,
                conv_flags=[h["conv_flag"] for h in hist], phi_first=state["phis"][0])
