"""Radiation tables: the INPUT format of the raytracing path.

Mirrors the interface of the reference's ``pyc2ray/radiation/common.py:11-35``
(``make_tau_table``) and ``pyc2ray/radiation/blackbody.py:25-110`` (``BlackBodySource``)
so callers of those can switch, but the quadrature is our own: fixed composite
Gauss-Legendre panels in frequency, vectorised over tau, instead of adaptive ``quad_vec``.

Table convention (reference ``common.py:25-34`` / ``photorates.f90:138-146``): length
NumTau+1, entry 0 is tau = 0, entries 1.. are tau = 10**(minlogtau + (i-1)*dlogtau).

Constants are hard-coded CODATA-2018 (cgs) -- astropy is not a dependency here.
"""
from __future__ import annotations

import numpy as np

__all__ = ["make_tau_table", "BlackBodySource", "H_PLANCK", "K_B", "C_LIGHT", "EV", "ION_FREQ_HI"]

H_PLANCK = 6.62607015e-27   # erg s
K_B = 1.380649e-16          # erg / K
C_LIGHT = 2.99792458e10     # cm / s
EV = 1.602176634e-12        # erg
ION_FREQ_HI = 3.2898419602508e15  # Rydberg frequency [Hz]


def make_tau_table(minlogtau, maxlogtau, NumTau):
    """Optical-depth grid of a C2Ray table; returns (tau[NumTau+1], dlogtau)."""
    dlogtau = (maxlogtau - minlogtau) / NumTau
    tau = np.empty(NumTau + 1)
    tau[0] = 0.0
    tau[1:] = 10 ** (minlogtau + np.arange(NumTau) * dlogtau)
    return tau, dlogtau


def _gl_nodes(f1, f2, panels=96, order=12):
    """Composite Gauss-Legendre nodes/weights on [f1, f2], panels uniform in log(freq)."""
    x, w = np.polynomial.legendre.leggauss(order)
    edges = np.exp(np.linspace(np.log(f1), np.log(f2), panels + 1))
    half = 0.5 * (edges[1:] - edges[:-1])
    mid = 0.5 * (edges[1:] + edges[:-1])
    nodes = (mid[:, None] + half[:, None] * x[None, :]).ravel()
    weights = (half[:, None] * w[None, :]).ravel()
    return nodes, weights


class BlackBodySource:
    """Black-body point source; same constructor and table methods as the reference class.

    ``pl_index`` is the power-law index of the cross-section frequency dependence
    ``(freq/freq0)**(-pl_index)``; ``grey=True`` makes the cross section constant.
    """

    def __init__(self, temp, grey, freq0, pl_index) -> None:
        self.temp = float(temp)
        self.grey = bool(grey)
        self.freq0 = float(freq0)
        self.pl_index = float(pl_index)
        self.R_star = 1.0

    # -- spectrum -------------------------------------------------------------------------
    def SED(self, freq):
        """Photon-number spectrum 4 pi R^2 * (2 pi / c^2) nu^2 / (exp(h nu / kT) - 1)."""
        freq = np.asarray(freq, dtype=np.float64)
        arg = freq * (H_PLANCK / K_B) / self.temp
        safe = np.minimum(arg, 700.0)
        sed = 4.0 * np.pi * self.R_star ** 2 * (2.0 * np.pi / C_LIGHT ** 2) * freq ** 2 / np.expm1(safe)
        return np.where(arg < 700.0, sed, 0.0)

    def integrate_SED(self, f1, f2):
        nu, w = _gl_nodes(f1, f2)
        return float(np.sum(w * self.SED(nu)))

    def normalize_SED(self, f1, f2, S_star_ref):
        scale = S_star_ref / self.integrate_SED(f1, f2)
        self.R_star = np.sqrt(scale) * self.R_star

    def cross_section_freq_dependence(self, freq):
        if self.grey:
            return np.ones_like(np.asarray(freq, dtype=np.float64))
        return (np.asarray(freq, dtype=np.float64) / self.freq0) ** (-self.pl_index)

    # -- tables ---------------------------------------------------------------------------
    def _tables(self, tau, f1, f2, S_star_ref, heat):
        self.normalize_SED(f1, f2, S_star_ref)
        tau = np.asarray(tau, dtype=np.float64)
        nu, w = _gl_nodes(f1, f2)
        s = self.cross_section_freq_dependence(nu)
        base = w * self.SED(nu)
        if heat:
            base = base * H_PLANCK * (nu - ION_FREQ_HI)
        thin = np.empty_like(tau)
        thick = np.empty_like(tau)
        step = 2048  # bound the (tau x nu) temporary
        for lo in range(0, tau.size, step):
            t = tau[lo:lo + step, None] * s[None, :]
            att = np.where(t < 700.0, np.exp(-np.minimum(t, 700.0)), 0.0)
            thick[lo:lo + step] = att @ base
            thin[lo:lo + step] = att @ (base * s)
        return thin, thick

    def make_photo_table(self, tau, freq_min, freq_max, S_star_ref):
        """(thin, thick) photo-ionisation tables: integrals of SED*s*exp(-tau s) and SED*exp(-tau s)."""
        return self._tables(tau, freq_min, freq_max, S_star_ref, heat=False)

    def make_heat_table(self, tau, freq_min, freq_max, S_star_ref):
        """(thin, thick) photo-heating tables: the photo integrands weighted by h (nu - nu_HI)."""
        return self._tables(tau, freq_min, freq_max, S_star_ref, heat=True)
