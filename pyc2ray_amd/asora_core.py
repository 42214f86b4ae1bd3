"""Mirror of ``pyc2ray/asora_core.py:19-65``: initialisation flag + guarded wrappers."""
from __future__ import annotations

from .load_extensions import load_asora

libasora = load_asora()

__all__ = ["cuda_is_init", "device_init", "device_close", "photo_table_to_device", "heat_table_to_device"]

cuda_init = False  # name kept from the reference; "device memory has been allocated"


def cuda_is_init():
    return cuda_init


def device_init(N, source_batch_size, rank, nr_gpus):
    """Initialise the GPU and allocate grid memory (asora_core.py:24-41)."""
    global cuda_init
    libasora.device_init(N, source_batch_size, rank, nr_gpus)
    cuda_init = True


def device_close():
    """Deallocate GPU memory (asora_core.py:44-53)."""
    global cuda_init
    if cuda_init:
        libasora.device_close()
        cuda_init = False
    else:
        raise RuntimeError("GPU not initialized. Please initialize it by calling device_init(N)")


def photo_table_to_device(thin_table, thick_table):
    """Copy radiation tables to the GPU; NumTau = table.shape[0] (asora_core.py:56-65)."""
    NumTau = thin_table.shape[0]
    if cuda_init:
        libasora.photo_table_to_device(thin_table, thick_table, NumTau)
    else:
        raise RuntimeError("GPU not initialized. Please initialize it by calling device_init(N)")


def heat_table_to_device(heat_thin_table, heat_thick_table):
    """Extension: heating tables for ``do_raytracing(..., heat=True)``."""
    NumTau = heat_thin_table.shape[0]
    if cuda_init:
        libasora.heat_table_to_device(heat_thin_table, heat_thick_table, NumTau)
    else:
        raise RuntimeError("GPU not initialized. Please initialize it by calling device_init(N)")
