"""The evolve loop of pyc2ray_amd.evolve driven by the CPU oracle -- TEST INFRASTRUCTURE.

Same host logic as the product (``pyc2ray_amd.evolve._evolve_loop``: iteration, convergence
criterion, relative-change tests), with the two device operations replaced by the oracle's
raytrace (one source per call, per-source flux, full box: the ASORA semantics)
and the oracle's ``global_pass``.  Used to generate tests/golden/evolve32.npz from the compiled
reference and to check the C restatement against it.

``evolve3D_host_sharded`` runs the product's WHOLE per-rank step logic (``_sharded_step``: criterion from
the global list, contiguous shard, all-reduce, collective decision) on the oracle, under whatever
``torch.distributed`` group the calling process has joined.
"""
import numpy as np

from pyc2ray_amd.evolve import _evolve_loop, _sharded_step


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


def evolve3D_host_sharded(oracle, c, history=None):
    """One step through ``pyc2ray_amd.evolve._sharded_step`` with the oracle as the device."""
    import torch
    N = c["N"]
    ndens, temp, xh, clump = (np.ascontiguousarray(c[k], dtype=np.float64) for k in ("ndens", "temp", "xh", "clump"))
    xh_av = xh.copy()
    xh_intermed = xh.copy()
    phi_t = torch.zeros(N ** 3, dtype=torch.float64)
    phi = phi_t.numpy().reshape(N, N, N)    # shares memory with the tensor that is all-reduced
    shard = {}

    def upload_sources(pos_flat, flux_flat, n_local):
        shard.update(pos=pos_flat.reshape(-1, 3), flux=flux_flat, n=n_local)

    def raytrace_local():
        if shard["n"] == 0:
            phi[...] = 0.0
            return
        o = oracle.do_all_sources(shard["flux"], shard["pos"], ndens, xh_av, c["sig"], c["dr"], c["thin"], c["thick"],
                                  c["minlogtau"], c["dlogtau"], c["R"], num_tau=c["num_tau"], flux_mode=1)
        phi[...] = o["phi_ion"]

    def chemistry():
        conv_flag = oracle.global_pass(c["dt"], ndens, temp, xh, xh_av, xh_intermed, phi, clump,
                                       c["bh00"], c["albpow"], c["colh0"], c["temph0"], c["abu_c"])
        return conv_flag, np.sum(xh_intermed), np.sum(1.0 - xh_intermed)

    hist = [] if history is None else history
    niter, n_local = _sharded_step(c["flux"], c["pos"].T, N ** 3, c["convergence_fraction"], upload_sources,
                                   raytrace_local, phi_t, chemistry, history=hist)
    return dict(xh_new=xh_intermed, xh_av=xh_av, phi_ion=phi.copy(), niter=niter, n_local=n_local,
                conv_flags=[h["conv_flag"] for h in hist], conv_criterion=hist[0]["conv_criterion"])
