"""Shared parity metric for phi_ion / phi_heat grids.

Bar (BASELINE.json north_star): <= 1e-10 relative per cell against the Fortran C2Ray raytracer.
A pure relative bound is not meaningful where the reference value itself is the result of a
cancellation T(tau_in) - T(tau_out) of nearly equal table values, or a sum over sources dominated
by another source; as SURVEY.md section 7 proposes the bound is
    |gpu - ref| <= RTOL * |ref| + AFLOOR * max|ref|
with RTOL = 1e-10 and AFLOOR = 1e-14, and the tests also report the worst pure-relative error and
the fraction of cells that needed the floor.
"""
import numpy as np

RTOL = 1e-10
AFLOOR = 1e-14


def compare(got, ref, rtol=RTOL, afloor=AFLOOR):
    got = np.asarray(got, dtype=np.float64).ravel()
    ref = np.asarray(ref, dtype=np.float64).ravel()
    scale = np.max(np.abs(ref)) if ref.size else 0.0
    err = np.abs(got - ref)
    tol_rel = rtol * np.abs(ref)
    ok = err <= tol_rel + afloor * scale
    nz = np.abs(ref) > 0
    rel = np.zeros_like(err)
    rel[nz] = err[nz] / np.abs(ref[nz])
    support_equal = bool(np.array_equal(got != 0, ref != 0))
    return dict(ok=bool(ok.all()), n_bad=int((~ok).sum()), max_rel=float(rel.max() if rel.size else 0.0),
                frac_floor=float(np.mean((err > tol_rel) & ok)) if err.size else 0.0,
                support_equal=support_equal, worst=int(np.argmax(rel)) if rel.size else -1,
                n_rel_gt=int((rel > rtol).sum()))
