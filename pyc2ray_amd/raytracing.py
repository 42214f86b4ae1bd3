"""Stand-alone raytracing call, GPU branch of the reference's ``pyc2ray/raytracing.py:36-163``."""
from __future__ import annotations

import numpy as np

from . import asora_core
from .sourceutils import format_sources

libasora = asora_core.libasora

__all__ = ["do_raytracing"]


def do_raytracing(dr, src_flux, src_pos, use_gpu, max_subbox, subboxsize, loss_fraction, ndens, xh_av,
                  photo_thin_table, photo_thick_table, heat_thin_table, heat_thick_table, minlogtau, dlogtau,
                  R_max_LLS, sig, logfile="pyC2Ray.log", quiet=True, stats=False, heat=False):
    """Same argument list as the reference.  Only the GPU branch exists here (``use_gpu`` must be
    true; the CPU branch of the reference is libc2ray, which is not part of this package).

    Returns ``(phi_ion, phi_heat)``; ``phi_heat`` is ``None`` unless ``heat=True`` (the reference's
    GPU branch returns an undefined name there, raytracing.py:163)."""
    if not use_gpu:
        raise NotImplementedError("pyc2ray_amd implements the ASORA (GPU) raytracing path only")
    if not asora_core.cuda_is_init():
        raise RuntimeError("GPU not initialized. Please initialize it by calling device_init(N)")
    NumSrc = src_flux.shape[0]
    N = ndens.shape[0]
    NumTau = photo_thin_table.shape[0]
    xh_av_flat = np.ravel(xh_av).astype("float64", copy=True)
    ndens_flat = np.ravel(ndens).astype("float64", copy=True)
    srcpos_flat, normflux_flat = format_sources(src_pos, src_flux)
    libasora.source_data_to_device(srcpos_flat, normflux_flat, NumSrc)
    coldensh_out_flat = np.ravel(np.zeros((N, N, N), dtype="float64"))
    phi_ion_flat = np.ravel(np.zeros((N, N, N), dtype="float64"))
    libasora.density_to_device(ndens_flat, N)
    if heat:
        phi_heat_flat = np.ravel(np.zeros((N, N, N), dtype="float64"))
        libasora.do_all_sources_heat(R_max_LLS, coldensh_out_flat, sig, dr, ndens_flat, xh_av_flat, phi_ion_flat,
                                     phi_heat_flat, NumSrc, N, minlogtau, dlogtau, NumTau)
        return np.reshape(phi_ion_flat, (N, N, N)), np.reshape(phi_heat_flat, (N, N, N))
    libasora.do_all_sources(R_max_LLS, coldensh_out_flat, sig, dr, ndens_flat, xh_av_flat, phi_ion_flat, NumSrc, N,
                            minlogtau, dlogtau, NumTau)
    return np.reshape(phi_ion_flat, (N, N, N)), None
