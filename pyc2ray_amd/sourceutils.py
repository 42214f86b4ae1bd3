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
