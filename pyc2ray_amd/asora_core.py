"""Host-side guard around the device library: the role of ``pyc2ray/asora_core.py:19-65``.

Same public names and error text as the reference (``cuda_is_init``, ``device_init``, ``device_close``,
``photo_table_to_device``; callers such as ``evolve3D`` test ``cuda_is_init()`` first), plus the heating-table
upload.  The state is one flag: "device_init has allocated the grids and device_close has not freed them".
"""
from __future__ import annotations

from .load_extensions import load_asora

libasora = load_asora()  # raises ImportError when the HIP library is not built: there is no CPU fallback

__all__ = ["cuda_is_init", "device_init", "device_close", "photo_table_to_device", "heat_table_to_device"]

_NOT_READY = "GPU not initialized. Please initialize it by calling device_init(N)"  # asora_core.py:51,63
_state = {"ready": False}


def cuda_is_init():
    return _state["ready"]


def _require_device():
    if not _state["ready"]:
        raise RuntimeError(_NOT_READY)


def device_init(N, source_batch_size, rank, nr_gpus):
    """Select GPU ``rank % nr_gpus`` and allocate the N^3 grids.  ``source_batch_size`` is accepted for
    compatibility: memory per source in flight is O(R^2) here, so there is nothing to size with it."""
    libasora.device_init(N, source_batch_size, rank, nr_gpus)
    _state["ready"] = True


def device_close():
    """Free the device memory; an error if nothing was allocated."""
    _require_device()
    libasora.device_close()
    _state["ready"] = False


def _upload(fn, thin, thick):
    _require_device()
    fn(thin, thick, thin.shape[0])  # NumTau is the table length, as in asora_core.py:59


def photo_table_to_device(thin_table, thick_table):
    """Upload the optically thin / thick photo-ionisation tables."""
    _upload(libasora.photo_table_to_device, thin_table, thick_table)


def heat_table_to_device(heat_thin_table, heat_thick_table):
    """Extension: upload the photo-heating tables used by ``do_raytracing(..., heat=True)``."""
    _upload(libasora.heat_table_to_device, heat_thin_table, heat_thick_table)
