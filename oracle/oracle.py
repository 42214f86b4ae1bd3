"""ctypes loader for the CPU oracle -- TEST INFRASTRUCTURE ONLY.

Two back ends with one calling convention:

* ``port``  -- oracle/libc2ray_oracle.so, our own C restatement (oracle/c2ray_oracle.c);
* ``port`` with ``log10="device"`` -- oracle/libc2ray_oracle_devlog10.so: the same restatement with ONE change,
  the table lookup's log10 is the host build of the HIP kernels' table-driven log10 (oracle/log10_device.cpp).
  It is the checker of the conditioning argument of tests/parity.py, never the parity reference;
* ``ref``   -- oracle/_ref/libc2ray_ref.so, the reference Fortran compiled by ``make -C oracle ref``
  in the build container (flang name mangling ``_QM<module>P<proc>``, every scalar by
  reference, trailing hidden sizes as in src/c2ray/raytracing.f90:52-56).

Only tests/, ``__graft_entry__.smoke()`` and bench.py's ``cpu_baseline`` leg import this
module.  The product package ``pyc2ray_amd`` never does.

All grids are logical (i, j, k) numpy arrays.  Source positions are 0-BASED (ASORA
convention); the +1 shift to the Fortran convention happens here.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
PORT_SO = os.path.join(HERE, "libc2ray_oracle.so")
DEVLOG_SO = os.path.join(HERE, "libc2ray_oracle_devlog10.so")
REF_SO = os.path.join(HERE, "_ref", "libc2ray_ref.so")

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def build_port(force: bool = False) -> str:
    src = os.path.join(HERE, "c2ray_oracle.c")
    if force or not os.path.exists(PORT_SO) or os.path.getmtime(PORT_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", HERE, "port"], stdout=subprocess.DEVNULL)
    return PORT_SO


def build_port_devlog10(force: bool = False) -> str:
    deps = [os.path.join(HERE, "c2ray_oracle.c"), os.path.join(HERE, "log10_device.cpp"),
            os.path.join(HERE, "..", "pyc2ray_amd", "csrc", "asora_log10.hpp")]
    if force or not os.path.exists(DEVLOG_SO) or any(os.path.getmtime(DEVLOG_SO) < os.path.getmtime(d) for d in deps):
        subprocess.check_call(["make", "-C", HERE, "port_devlog10"], stdout=subprocess.DEVNULL)
    return DEVLOG_SO


def build_ref(reference: str = "/root/reference") -> str | None:
    """Compile the reference Fortran where it lies (build container only)."""
    if not os.path.isdir(os.path.join(reference, "src", "c2ray")):
        return REF_SO if os.path.exists(REF_SO) else None
    subprocess.check_call(["make", "-C", HERE, "ref", f"REFERENCE={reference}"], stdout=subprocess.DEVNULL)
    return REF_SO


def _d(a):
    return a.ctypes.data_as(_dp)


class PortOracle:
    """The C restatement.  Grids are C-ordered float64 (N,N,N)."""

    kind = "port"

    def __init__(self, log10: str = "libm"):
        """log10="device": the variant whose lookups use the HIP kernels' log10 (see the module docstring)."""
        if log10 not in ("libm", "device"):
            raise ValueError(log10)
        self.lib = C.CDLL(build_port() if log10 == "libm" else build_port_devlog10())
        self.log10 = log10
        L = self.lib
        L.c2ray_oracle_log10_kind.restype = C.c_int
        assert L.c2ray_oracle_log10_kind() == (1 if log10 == "device" else 0)
        L.c2ray_do_all_sources.restype = C.c_longlong
        L.c2ray_do_all_sources.argtypes = [
            _dp, _ip, C.c_int, C.c_int, _dp, C.c_double, C.c_double, _dp, _dp, _dp, _dp, _ip, _dp, C.c_float,
            _dp, _dp, _dp, _dp, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
            C.c_int]
        L.c2ray_cinterp.restype = None
        L.c2ray_cinterp.argtypes = [_ip, _ip, _dp, _dp, _dp, C.c_double, C.c_int, C.c_int, C.c_int]
        L.c2ray_global_pass.restype = C.c_int
        L.c2ray_global_pass.argtypes = [C.c_double, _dp, _dp, _dp, _dp, _dp, _dp, _dp, C.c_double, C.c_double,
                                        C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int]
        L.c2ray_photoion_rates.restype = None
        L.c2ray_photoion_rates.argtypes = [C.c_double] * 5 + [_dp, _dp, _dp, _dp, _dp, _dp, _dp, C.c_double,
                                                               C.c_double, C.c_int]
        L.c2ray_set_sens_outputs.restype = None
        L.c2ray_set_sens_outputs.argtypes = [_dp, _dp, C.c_longlong]
        L.c2ray_asora_visit_count.restype = None
        L.c2ray_asora_visit_count.argtypes = [C.c_double, C.c_int, C.POINTER(C.c_longlong),
                                              C.POINTER(C.c_longlong)]

    # -- raytracing -------------------------------------------------------------------
    def do_all_sources(self, flux, pos0, ndens, xh_av, sig, dr, thin, thick, minlogtau, dlogtau, R,
                       heat_thin=None, heat_thick=None, max_subbox=None, subboxsize=None, loss_fraction=0.0,
                       flux_mode=1, num_tau=None, sens=False):
        """pos0: (Ns,3) 0-based.  sens=True also returns phi_sens / heat_sens: per cell, the summed change of
        each source's rate under a +-1 ulp change of the libm log10 results (conditioning of the
        reference value itself; see tests/parity.py).  Returns dict(phi_ion, phi_heat, coldensh_out, nsubbox, photon_loss, visits)."""
        N = ndens.shape[0]
        flux = np.ascontiguousarray(flux, dtype=np.float64)
        pos1 = np.ascontiguousarray(np.asarray(pos0, dtype=np.int32).reshape(-1, 3) + 1)  # [ns][d] == (3,Ns) F-order
        Ns = flux.shape[0]
        nd = np.ascontiguousarray(ndens, dtype=np.float64)
        xh = np.ascontiguousarray(xh_av, dtype=np.float64)
        thin = np.ascontiguousarray(thin, dtype=np.float64)
        thick = np.ascontiguousarray(thick, dtype=np.float64)
        ht = np.zeros_like(thin) if heat_thin is None else np.ascontiguousarray(heat_thin, dtype=np.float64)
        hk = np.zeros_like(thick) if heat_thick is None else np.ascontiguousarray(heat_thick, dtype=np.float64)
        NumTau = thin.shape[0] if num_tau is None else int(num_tau)
        if max_subbox is None:
            max_subbox = N
        if subboxsize is None:
            subboxsize = N
        cdh = np.zeros((N, N, N))
        phi = np.zeros((N, N, N))
        heat = np.zeros((N, N, N))
        nbox = C.c_int(0)
        loss = C.c_double(0.0)
        psens = hsens = None
        if sens:
            psens, hsens = np.zeros((N, N, N)), np.zeros((N, N, N))
            self.lib.c2ray_set_sens_outputs(_d(psens), _d(hsens), N ** 3)
        else:
            self.lib.c2ray_set_sens_outputs(None, None, 0)
        visits = self.lib.c2ray_do_all_sources(
            _d(flux), pos1.ctypes.data_as(_ip), int(max_subbox), int(subboxsize), _d(cdh), float(sig), float(dr),
            _d(nd), _d(xh), _d(phi), _d(heat), C.byref(nbox), C.byref(loss), float(loss_fraction), _d(thin),
            _d(thick), _d(ht), _d(hk), float(minlogtau), float(dlogtau), float(R), NumTau, Ns, N, N, N,
            int(flux_mode))
        self.lib.c2ray_set_sens_outputs(None, None, 0)
        return dict(phi_ion=phi, phi_heat=heat, coldensh_out=cdh, nsubbox=nbox.value, photon_loss=loss.value,
                    visits=int(visits), phi_sens=psens, heat_sens=hsens)

    def cinterp(self, pos0, srcpos0, cdh, sigma):
        """0-based positions (pos may lie outside the box: it is wrapped like the reference does)."""
        N = cdh.shape[0]
        cd = np.ascontiguousarray(cdh, dtype=np.float64)
        p = (C.c_int * 3)(*[int(v) + 1 for v in pos0])
        s = (C.c_int * 3)(*[int(v) + 1 for v in srcpos0])
        a, b = C.c_double(), C.c_double()
        self.lib.c2ray_cinterp(p, s, C.byref(a), C.byref(b), _d(cd), float(sigma), N, N, N)
        return a.value, b.value

    # -- chemistry --------------------------------------------------------------------
    def global_pass(self, dt, ndens, temp, xh, xh_av, xh_intermed, phi_ion, clump, bh00, albpow, colh0, temph0,
                    abu_c):
        """Updates xh_av and xh_intermed IN PLACE (C-ordered float64) and returns conv_flag."""
        N = ndens.shape[0]
        for a in (xh_av, xh_intermed):
            assert a.flags.c_contiguous and a.dtype == np.float64
        g = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        nd, tp, x0, ph, cl = g(ndens), g(temp), g(xh), g(phi_ion), g(clump)
        return int(self.lib.c2ray_global_pass(float(dt), _d(nd), _d(tp), _d(x0), _d(xh_av), _d(xh_intermed),
                                              _d(ph), _d(cl), float(bh00), float(albpow), float(colh0),
                                              float(temph0), float(abu_c), N, N, N))

    def asora_visit_count(self, R, N):
        v, s = C.c_longlong(), C.c_longlong()
        self.lib.c2ray_asora_visit_count(float(R), int(N), C.byref(v), C.byref(s))
        return v.value, s.value


class RefOracle:
    """The reference Fortran itself (build container; prebuilt .so on the GPU box)."""

    kind = "reference"

    def __init__(self, path: str | None = None):
        path = path or REF_SO
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        self.lib = C.CDLL(path)

    def do_all_sources(self, flux, pos0, ndens, xh_av, sig, dr, thin, thick, minlogtau, dlogtau, R,
                       heat_thin=None, heat_thick=None, max_subbox=None, subboxsize=None, loss_fraction=0.0,
                       flux_mode=1, num_tau=None):
        """Same signature as PortOracle.do_all_sources.

        flux_mode=1 (per-source flux) is realised by calling the reference ONE SOURCE PER CALL and
        summing phi in source order; flux_mode=0 is a single call.
        """
        N = ndens.shape[0]
        flux = np.ascontiguousarray(flux, dtype=np.float64)
        pos1 = np.asarray(pos0, dtype=np.int32).reshape(-1, 3) + 1
        Ns = flux.shape[0]
        F = lambda a: np.asfortranarray(a, dtype=np.float64)
        nd, xh = F(ndens), F(xh_av)
        thin = np.ascontiguousarray(thin, dtype=np.float64)
        thick = np.ascontiguousarray(thick, dtype=np.float64)
        ht = np.zeros_like(thin) if heat_thin is None else np.ascontiguousarray(heat_thin, dtype=np.float64)
        hk = np.zeros_like(thick) if heat_thick is None else np.ascontiguousarray(heat_thick, dtype=np.float64)
        NumTau = thin.shape[0] if num_tau is None else int(num_tau)
        # The reference can read table(NumTau+2) -- one element past the end (photorates.f90:140-146,
        # SURVEY.md 3.4 #9).  Give that read a defined value (= last entry) instead of heap garbage.
        pad = lambda t: np.concatenate([t[:NumTau], np.full(NumTau + 2 - min(NumTau, t.shape[0]), t[min(NumTau, t.shape[0]) - 1])])
        thin, thick, ht, hk = pad(thin), pad(thick), pad(ht), pad(hk)
        if max_subbox is None:
            max_subbox = N
        if subboxsize is None:
            subboxsize = N
        fn = self.lib._QMraytracingPdo_all_sources
        fn.restype = None
        phi_tot = np.zeros((N, N, N), order="F")
        heat_tot = np.zeros((N, N, N), order="F")
        cdh = np.zeros((N, N, N), order="F")
        nbox_tot, loss_tot = 0, 0.0
        groups = [(i, i + 1) for i in range(Ns)] if flux_mode == 1 else [(0, Ns)]
        bi = lambda v: C.byref(C.c_int(int(v)))
        bd = lambda v: C.byref(C.c_double(float(v)))
        for lo, hi in groups:
            fl = np.ascontiguousarray(flux[lo:hi])
            ps = np.asfortranarray(pos1[lo:hi].T.astype(np.int32))  # (3, n)
            phi = np.zeros((N, N, N), order="F")
            heat = np.zeros((N, N, N), order="F")
            nbox = C.c_int(0)
            loss = C.c_double(0.0)
            fn(_d(fl), ps.ctypes.data_as(_ip), bi(max_subbox), bi(subboxsize), _d(cdh), bd(sig), bd(dr), _d(nd),
               _d(xh), _d(phi), _d(heat), C.byref(nbox), C.byref(loss), C.byref(C.c_float(loss_fraction)),
               _d(thin), _d(thick), _d(ht), _d(hk), bd(minlogtau), bd(dlogtau), bd(R), bi(NumTau), bi(hi - lo),
               bi(N), bi(N), bi(N))
            phi_tot = phi_tot + phi  # source order
            heat_tot = heat_tot + heat
            nbox_tot += nbox.value
            loss_tot += loss.value
        return dict(phi_ion=np.ascontiguousarray(phi_tot), phi_heat=np.ascontiguousarray(heat_tot),
                    coldensh_out=np.ascontiguousarray(cdh), nsubbox=nbox_tot, photon_loss=loss_tot, visits=None)

    def cinterp(self, pos0, srcpos0, cdh, sigma):
        N = cdh.shape[0]
        cd = np.asfortranarray(cdh, dtype=np.float64)
        p = (C.c_int * 3)(*[int(v) + 1 for v in pos0])
        s = (C.c_int * 3)(*[int(v) + 1 for v in srcpos0])
        a, b = C.c_double(), C.c_double()
        n = C.c_int(N)
        fn = self.lib._QMraytracingPcinterp
        fn.restype = None
        fn(p, s, C.byref(a), C.byref(b), _d(cd), C.byref(C.c_double(float(sigma))), C.byref(n), C.byref(n),
           C.byref(n))
        return a.value, b.value

    def global_pass(self, dt, ndens, temp, xh, xh_av, xh_intermed, phi_ion, clump, bh00, albpow, colh0, temph0,
                    abu_c):
        """Updates xh_av and xh_intermed in place (any layout) and returns conv_flag."""
        N = ndens.shape[0]
        F = lambda a: np.asfortranarray(a, dtype=np.float64)
        nd, tp, x0, ph, cl = F(ndens), F(temp), F(xh), F(phi_ion), F(clump)
        xa, xi = F(xh_av).copy(order="F"), F(xh_intermed).copy(order="F")
        conv = C.c_int(0)
        n = C.c_int(N)
        bd = lambda v: C.byref(C.c_double(float(v)))
        fn = self.lib._QMchemistryPglobal_pass
        fn.restype = None
        fn(bd(dt), _d(nd), _d(tp), _d(x0), _d(xa), _d(xi), _d(ph), _d(cl), bd(bh00), bd(albpow), bd(colh0),
           bd(temph0), bd(abu_c), C.byref(conv), C.byref(n), C.byref(n), C.byref(n))
        xh_av[...] = xa
        xh_intermed[...] = xi
        return conv.value


def get_oracle(prefer: str = "port"):
    """prefer='ref' returns the compiled reference when its .so exists, else the port."""
    if prefer == "ref" and os.path.exists(REF_SO):
        return RefOracle()
    return PortOracle()
