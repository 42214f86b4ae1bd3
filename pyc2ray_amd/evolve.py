"""One evolve step of the ionised fraction -- the caller of the raytracing path.

Mirror of the reference's ``pyc2ray/evolve.py:41-416`` (GPU branch): iterate
``raytrace -> sum over ranks -> chemistry -> convergence test`` until the time-averaged ionised
fraction has converged, then return ``(xh_new, phi_ion, coldensh_out)``.

What is kept from the reference: the argument list, the flattening (C-order ``ravel`` +
``astype(float64)``, ``:171-172``), the source wire format and contiguous sharding (``:173-189``), one
upload of sources and density per step (``:192,201``), the convergence criterion
``conv_criterion = min(int(convergence_fraction*N^3), (NumSrc-1)/3)`` (``:154``) -- formed from the
**global** source count of ``:147``: the reference overwrites ``NumSrc`` with the shard size only
afterwards, at ``:183`` -- and the two relative-change tests (``:350-373``).

What is different (MI355X-first): all grids stay on the device for the whole step.  The raytrace
writes phi_ion into a device tensor, the rank sum is ONE in-place RCCL all-reduce of that tensor
(replaces the host ``Reduce`` + ``Bcast``, ``:309-317``), the chemistry pass is a HIP kernel on the
same tensors (replaces the rank-0 Fortran ``global_pass`` and the three N^3 broadcasts that
followed it, ``:326-340, :386-398``) and also returns the two sums of the convergence test.  Per
iteration 24 bytes cross PCIe.  Every rank runs the (deterministic, per-cell) chemistry on
identical inputs; the convergence decision is taken on rank 0 and broadcast, so ranks cannot diverge
on it.

``_evolve_loop`` is the iteration with the two device operations injected; ``_sharded_step`` is the
whole host logic of a step (criterion from the global list, shard, upload, loop, collective decision)
with every device operation injected.  ``evolve3D`` runs it on the HIP library; the tests run the very
same function on the CPU oracle, single rank and world-2 gloo (tests/test_evolve_host.py,
tests/test_distributed_cpu.py).
"""
from __future__ import annotations

import numpy as np

from .sourceutils import format_sources, shard_sources

__all__ = ["evolve3D", "conv_criterion"]


def conv_criterion(convergence_fraction, num_cells, num_src):
    """evolve.py:154 -- ``num_src`` is the length of the whole source list (``:147``), on every rank."""
    return min(int(convergence_fraction * num_cells), (num_src - 1) / 3)


def _evolve_loop(raytrace, chemistry, num_cells, num_src, convergence_fraction, decide=None, max_iter=10000,
                 history=None):
    """The ``while not converged`` loop of evolve.py:232-401.

    ``raytrace()`` must leave the all-rank phi_ion where ``chemistry`` finds it; ``chemistry()``
    returns ``(conv_flag, sum(xh_intermed), sum(1 - xh_intermed))``.  ``num_src`` is the GLOBAL number of
    sources (see ``conv_criterion``).  ``decide(bool) -> bool`` makes the decision collective (rank 0's
    value wins); identity for a single process.  Returns the number of iterations."""
    crit = conv_criterion(convergence_fraction, num_cells, num_src)
    prev_sum_xh1_int = 2 * num_cells
    prev_sum_xh0_int = 2 * num_cells
    converged = False
    niter = 0
    while not converged:
        niter += 1
        raytrace()
        conv_flag, sum_xh1_int, sum_xh0_int = chemistry()
        if sum_xh1_int > 0.0:
            rel_change_xh1 = abs((sum_xh1_int - prev_sum_xh1_int) / sum_xh1_int)
        else:
            rel_change_xh1 = 1.0
        if sum_xh0_int > 0.0:
            rel_change_xh0 = abs((sum_xh0_int - prev_sum_xh0_int) / sum_xh0_int)
        else:
            rel_change_xh0 = 1.0
        converged = (conv_flag < crit) or ((rel_change_xh1 < convergence_fraction)
                                           and (rel_change_xh0 < convergence_fraction))
        if decide is not None:
            converged = decide(converged)
        if history is not None:
            history.append(dict(conv_flag=int(conv_flag), rel_change_xh1=float(rel_change_xh1),
                                rel_change_xh0=float(rel_change_xh0), conv_criterion=float(crit)))
        prev_sum_xh1_int = sum_xh1_int
        prev_sum_xh0_int = sum_xh0_int
        if niter >= max_iter:
            raise RuntimeError(f"evolve3D: no convergence after {niter} iterations")
    return niter


def _process_group():
    """(rank, nprocs, distributed) of ``torch.distributed``'s default group; (0, 1, False) without one."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return dist.get_rank(), dist.get_world_size(), True
    return 0, 1, False


def _sharded_step(src_flux, src_pos, num_cells, convergence_fraction, upload_sources, raytrace_local, rate_grid,
                  chemistry, device="cpu", history=None):
    """Host logic of one evolve step on this rank (evolve.py:146-192, 232-401).

    ``src_flux`` (Ns,), ``src_pos`` (3, Ns): the WHOLE list, identical on every rank.
    ``upload_sources(pos_flat, flux_flat, n_local)`` gets this rank's contiguous shard once;
    ``raytrace_local()`` overwrites ``rate_grid`` (a torch tensor on ``device``) with the shard's rates;
    the grid is then summed over the ranks in place; ``chemistry()`` as in ``_evolve_loop``.
    Returns ``(niter, n_local)``."""
    import torch
    import torch.distributed as dist
    rank, nprocs, distributed = _process_group()
    src_flux = np.asarray(src_flux)
    src_pos = np.asarray(src_pos)
    num_src_global = src_flux.shape[0]          # evolve.py:147 -- the criterion of :154 is formed from this
    lo, hi = shard_sources(num_src_global, rank, nprocs)
    n_local = hi - lo                           # evolve.py:183
    upload_sources(*format_sources(src_pos[:, lo:hi], src_flux[lo:hi]), n_local)

    def raytrace():
        raytrace_local()
        if distributed:
            dist.all_reduce(rate_grid)          # replaces Reduce + Bcast, evolve.py:309-317

    def decide(flag):
        if not distributed:
            return flag
        t = torch.tensor([int(flag)], dtype=torch.int32, device=device)
        dist.broadcast(t, src=0)                # evolve.py:398
        return bool(t.item())

    niter = _evolve_loop(raytrace, chemistry, num_cells, num_src_global, convergence_fraction, decide,
                         history=history)
    return niter, n_local


def evolve3D(dt, dr, src_flux, src_pos, use_gpu, max_subbox, subboxsize, loss_fraction, use_mpi, comm, rank,
             nprocs, temp, ndens, xh, clump, photo_thin_table, photo_thick_table, minlogtau, dlogtau, R_max_LLS,
             convergence_fraction, sig, bh00, albpow, colh0, temph0, abu_c, logfile="pyC2Ray.log", quiet=True,
             history=None, want_coldens=True, sweep_unobserved=False):
    """Argument list of the reference (evolve.py:41-72).  ``use_gpu`` must be true.  ``use_mpi`` /
    ``comm`` are ignored: the process group is ``torch.distributed``'s (if initialised), ``rank`` /
    ``nprocs`` are taken from it.  Tables must have been uploaded with ``photo_table_to_device``.

    The cells outside a source's R-sphere are never upwind of a cell inside it: all they feed is the exported
    column density, which is the LAST source's.  The step therefore sweeps them for that source alone (library
    option ``skip_unobserved``, put back to the caller's value afterwards): same three grids, about a quarter less
    raytracing time.  ``sweep_unobserved=True`` sweeps them for every source, as a stand-alone ``do_all_sources`` does.

    Returns ``(xh_new, phi_ion, coldensh_out)`` as (N, N, N) float64 numpy arrays."""
    import torch

    from . import asora_core
    libasora = asora_core.libasora
    if not use_gpu:
        raise NotImplementedError("pyc2ray_amd implements the ASORA (GPU) branch of evolve3D only")
    if not asora_core.cuda_is_init():
        raise RuntimeError("GPU not initialized. Please initialize it by calling device_init(N)")

    N = temp.shape[0]
    NumCells = N * N * N
    NumTau = photo_thin_table.shape[0]

    # the library's grids live on the device device_init selected: the step's tensors must live there too
    dev = torch.device("cuda", libasora.device_index())
    up = lambda a: torch.from_numpy(np.ravel(a).astype("float64", copy=True)).to(dev)
    ndens_t, temp_t, xh_t, clump_t = up(ndens), up(temp), up(xh), up(clump)
    xh_av_t = xh_t.clone()
    xh_intermed_t = xh_t.clone()
    phi_t = torch.zeros(NumCells, dtype=torch.float64, device=dev)
    cdh_t = torch.zeros(NumCells, dtype=torch.float64, device=dev) if want_coldens else None
    stream = torch.cuda.current_stream(dev).cuda_stream
    libasora.density_from_device(ndens_t.data_ptr(), N, stream)
    local = {}

    def upload_sources(pos_flat, flux_flat, n_local):
        local["n"] = n_local
        if n_local:
            libasora.source_data_to_device(pos_flat, flux_flat, n_local)

    def raytrace_local():
        if local["n"] == 0:        # more ranks than sources: this rank adds nothing to the sum
            phi_t.zero_()
            return
        libasora.do_all_sources_dev(R_max_LLS, cdh_t.data_ptr() if want_coldens else 0, sig, dr, xh_av_t.data_ptr(),
                                    phi_t.data_ptr(), 0, local["n"], N, minlogtau, dlogtau, NumTau, stream)

    def chemistry():
        return libasora.global_pass_dev(dt, ndens_t.data_ptr(), temp_t.data_ptr(), xh_t.data_ptr(),
                                        xh_av_t.data_ptr(), xh_intermed_t.data_ptr(), phi_t.data_ptr(),
                                        clump_t.data_ptr(), bh00, albpow, colh0, temph0, abu_c, N, stream)

    skip_before = libasora.get_option("skip_unobserved")
    if not sweep_unobserved:
        libasora.set_option("skip_unobserved", 1)
    try:
        _sharded_step(src_flux, src_pos, NumCells, convergence_fraction, upload_sources, raytrace_local, phi_t,
                      chemistry, device=dev, history=history)
    finally:
        libasora.set_option("skip_unobserved", skip_before)
    shape = (N, N, N)
    xh_new = xh_intermed_t.cpu().numpy().reshape(shape)
    phi_ion = phi_t.cpu().numpy().reshape(shape)
    coldensh_out = cdh_t.cpu().numpy().reshape(shape) if want_coldens else None
    return xh_new, phi_ion, coldensh_out
