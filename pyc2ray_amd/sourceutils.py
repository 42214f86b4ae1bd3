"""Source-list wire format of the path (reference: ``pyc2ray/utils/sourceutils.py:7-33``)."""
from __future__ import annotations

import numpy as np


def format_sources(source_pos, source_flux):
    """(3, Ns) positions -> flat int32 x0,y0,z0,x1,... (column-major ravel); flux -> float64.

    Positions are used as given (0-based for ASORA; no shift, like the reference)."""
    source_pos = np.asarray(source_pos)
    if source_pos.ndim != 2 or source_pos.shape[0] != 3:
        raise ValueError("source_pos must have shape (3, numsrc)")
    source_pos_flat = np.ascontiguousarray(np.ravel(source_pos.astype("int32"), order="F"))
    source_flux_flat = np.ascontiguousarray(np.asarray(source_flux).astype("float64"))
    return source_pos_flat, source_flux_flat


def shard_sources(num_src, rank, nprocs):
    """Contiguous source shard of ``rank``: ``perrank = Ns // nprocs``, the last rank takes the
    remainder (reference: ``pyc2ray/evolve.py:175-180``).  Returns (i_start, i_end)."""
    perrank = num_src // nprocs
    i_start = int(rank * perrank)
    i_end = int((rank + 1) * perrank) if rank != nprocs - 1 else num_src
    return i_start, i_end


def merge_sources(source_pos, source_flux):
    """Sum the fluxes of sources that share a cell.  ``source_pos`` (3, Ns) integer grid positions.
    Halo catalogues put many sources into one cell; the raytrace is linear in the flux of a cell, so
    one sweep per occupied cell gives the same rates (to rounding) at a fraction of the cost.  Returns
    ``(pos (3, Nu), flux (Nu,))`` ordered by flat cell index."""
    source_pos = np.asarray(source_pos)
    source_flux = np.asarray(source_flux, dtype=np.float64)
    if source_pos.ndim != 2 or source_pos.shape[0] != 3:
        raise ValueError("source_pos must have shape (3, numsrc)")
    if source_pos.shape[1] == 0:
        return source_pos.astype(np.int32), source_flux
    uniq, inv = np.unique(source_pos.astype(np.int64), axis=1, return_inverse=True)
    flux = np.bincount(np.ravel(inv), weights=source_flux, minlength=uniq.shape[1])
    return uniq.astype(np.int32), flux


def bin_sources(srcpos_mpc, mstar_msun, boxsize, meshsize):
    """Bin source positions (Ns, 3) in box units onto the mesh and sum their masses per cell
    (reference: ``pyc2ray/utils/other_utils.py:145-158``, which uses scipy's ``binned_statistic_dd``;
    bins are [k*dx, (k+1)*dx), the last one closed).  Returns ``(srcpos (Nu, 3) int, srcmstar (Nu,))``
    in ``np.argwhere`` (C) order, cells with zero summed mass dropped."""
    pos = np.asarray(srcpos_mpc, dtype=np.float64)
    mass = np.asarray(mstar_msun, dtype=np.float64)
    edges = np.linspace(0, boxsize, meshsize + 1)
    idx = np.stack([np.clip(np.searchsorted(edges, pos[:, d], side="right") - 1, -1, meshsize) for d in range(3)], axis=1)
    on_top = pos == boxsize  # the last bin is closed on the right
    idx[on_top] = meshsize - 1
    inside = np.all((idx >= 0) & (idx < meshsize), axis=1)
    flat = np.ravel_multi_index(idx[inside].T, (meshsize,) * 3)
    summed = np.bincount(flat, weights=mass[inside], minlength=meshsize ** 3)
    keep = np.flatnonzero(summed > 0)
    return np.stack(np.unravel_index(keep, (meshsize,) * 3), axis=1), summed[keep]
