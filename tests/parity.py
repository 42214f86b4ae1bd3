"""Shared parity metric for phi_ion / phi_heat grids.

Bar (BASELINE.json north_star): <= 1e-10 relative per cell against the Fortran C2Ray raytracer.
A pure relative bound is not meaningful where the reference value itself is the result of a
cancellation T(tau_in) - T(tau_out) of nearly equal table values, or a sum over sources dominated
by another source; as SURVEY.md section 7 proposes the bound is
    |gpu - ref| <= RTOL * |ref| + AFLOOR * max|ref|
with RTOL = 1e-10 and AFLOOR = 1e-14, and the tests also report the worst pure-relative error and
the fraction of cells that needed the floor.

Conditioning term.  In a "thick" cell the rate is prefact*(T(tau_in) - T(tau_out)) with both T's read
from a table at index 1 + (log10(tau) - minlogtau)/dlogtau.  When tau_out/tau_in - 1 is tiny (highly
ionised cells behind an optical depth) the difference cancels up to 7 digits, and the REFERENCE
value itself moves by up to ~1e-5 relative when its libm's log10 result moves by one ulp (glibc's
log10 is not correctly rounded, so this is ordinary libm-to-libm variation).  No implementation
with a different log10 can match such a cell to 1e-10, and the mismatch carries no information.
The oracle therefore measures, per cell, `sens` = the summed change of every source's rate under a
+-1 ulp change of its log10 results (oracle/c2ray_oracle.c: c2ray_photoion_rates_sens; stored in the
goldens as phi_sens), and the bound becomes
    |gpu - ref| <= RTOL*|ref| + AFLOOR*max|ref| + KSENS*sens,   KSENS = 4
(device log10 within 1-2 ulp of glibc's, two lookups per cell).  Cells that pass only through the
sens term are counted and printed (`n_sens`); for all well-conditioned inputs that count is 0.
"""
import numpy as np

RTOL = 1e-10
AFLOOR = 1e-14
KSENS = 4.0


def compare(got, ref, sens=None, rtol=RTOL, afloor=AFLOOR, ksens=KSENS):
    got = np.asarray(got, dtype=np.float64).ravel()
    ref = np.asarray(ref, dtype=np.float64).ravel()
    scale = np.max(np.abs(ref)) if ref.size else 0.0
    err = np.abs(got - ref)
    tol_rel = rtol * np.abs(ref)
    ok_plain = err <= tol_rel + afloor * scale
    ok = ok_plain
    if sens is not None:
        ok = err <= tol_rel + afloor * scale + ksens * np.asarray(sens, dtype=np.float64).ravel()
    nz = np.abs(ref) > 0
    rel = np.zeros_like(err)
    rel[nz] = err[nz] / np.abs(ref[nz])
    support_equal = bool(np.array_equal(got != 0, ref != 0))
    return dict(ok=bool(ok.all()), n_bad=int((~ok).sum()), max_rel=float(rel.max() if rel.size else 0.0),
                frac_floor=float(np.mean((err > tol_rel) & ok)) if err.size else 0.0,
                support_equal=support_equal, worst=int(np.argmax(rel)) if rel.size else -1,
                n_sens=int((ok & ~ok_plain).sum()),
                n_rel_gt=int((rel > rtol).sum()))


def record(config, r, **extra):
    """Append one line {config, max_rel, n_rel_gt, frac_floor, n_sens, support_equal, ...} to the parity report
    the run leaves behind ($ASORA_PARITY_REPORT, default gpurun_out/parity_report.jsonl at the repo root; the
    committed copy of a round's GPU run is profiles/rN_parity_report.json)."""
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.environ.get("ASORA_PARITY_REPORT") or os.path.join(root, "gpurun_out", "parity_report.jsonl")
    line = dict(config=config, **{k: r[k] for k in ("max_rel", "n_rel_gt", "frac_floor", "n_sens", "n_bad",
                                                    "support_equal", "ok")}, **extra)
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "a") as f:
            f.write(json.dumps(line) + "\n")
    except OSError:
        pass  # a read-only tree must not fail a parity test
    return line
