#!/usr/bin/env python3
"""Wall time of one device-resident evolve step (raytrace -> chemistry -> convergence, pyc2ray_amd.evolve.evolve3D)
on the bench inputs: 256^3 uniform grid, R = 60, NS random sources, dt = 1 Myr."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases  # noqa: E402
from pyc2ray_amd import asora_core  # noqa: E402
from pyc2ray_amd.evolve import evolve3D  # noqa: E402

N = 256
ns = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
rng = np.random.default_rng(918)
pos = rng.integers(0, N, (ns, 3)).astype(np.int32)
flux = 4.09e40 * rng.uniform(1e9, 5e11, ns) / 1e48
c = cases._base(N, 60.0, 3.0 * cases.MPC / 200, pos, flux, xh=np.full((N, N, N), 2e-4),
                table_kw=dict(temp=1e5, numtau=20000))
c.update(cases.CHEM)
asora_core.device_init(N, 64, 0, 1)
asora_core.photo_table_to_device(c["thin"], c["thick"])
temp, clump = np.full((N, N, N), 1e4), np.ones((N, N, N))
out = {}
# the step's default (cells outside the R-sphere for the last source only), then every cell for every source
for mode, sweep_unobserved in (("default", False), ("sweep_unobserved", True)):
    for rep in range(2):
        hist = []
        t0 = time.perf_counter()
        xh_new, phi, _ = evolve3D(1e6 * cases.YEAR, c["dr"], flux, pos.T, True, 0, 0, 0.0, False, None, 0, 1, temp,
                                  c["ndens"], c["xh"], clump, c["thin"], c["thick"], c["minlogtau"], c["dlogtau"], c["R"],
                                  1e-4, c["sig"], *[c[k] for k in ("bh00", "albpow", "colh0", "temph0", "abu_c")],
                                  history=hist, want_coldens=False, sweep_unobserved=sweep_unobserved)
        dt = time.perf_counter() - t0
        out[mode] = dict(sources=ns, iterations=len(hist), seconds=dt, ms_per_iteration=dt / len(hist) * 1e3,
                         mean_xHII=float(xh_new.mean()))
asora_core.device_close()
print(json.dumps(out))
