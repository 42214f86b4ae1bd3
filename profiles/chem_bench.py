#!/usr/bin/env python3
"""Times the chemistry pass (asora_global_pass_dev) on device-resident 256^3 grids: chem_bench.py [N] [isothermal].
Algorithmic bytes: 5 streams read + 2 read-modify-written = 72 B per cell."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases  # noqa: E402
from pyc2ray_amd import libasora as asora  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
c = cases.chemistry_case(N=N, seed=1)
if len(sys.argv) > 2 and sys.argv[2] == "isothermal":  # what every driver of the reference passes: temp0 * ones
    c["temp"] = np.full_like(c["temp"], 1.0e4)
keys = ("ndens", "temp", "xh", "xh_av", "xh_intermed", "phi_ion", "clump")
dev = torch.device("cuda", 0)
t0 = {k: torch.from_numpy(c[k].ravel().copy()).to(dev) for k in keys}
chem = [c[k] for k in ("bh00", "albpow", "colh0", "temph0", "abu_c")]
asora.device_init(N, 8, 0, 1)
stream = torch.cuda.current_stream().cuda_stream
res = {}
for label, dt in (("hard (random cells, many inner iterations)", c["dt"]), ("easy (dt -> 0, one inner iteration)", 1e3)):
    ms = []
    for it in range(6):
        t = {k: v.clone() for k, v in t0.items()}
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        asora.global_pass_dev(dt, *[t[k].data_ptr() for k in keys], *chem, N, stream, want_stats=False)
        e1.record()
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    best = float(np.median(ms[1:]))
    res[label] = dict(ms=best, GBps=72.0 * N ** 3 / best / 1e6, frac_of_8TBps=72.0 * N ** 3 / best / 1e6 / 8000.0)
asora.device_close()
print(json.dumps(dict(N=N, temperature=sys.argv[2] if len(sys.argv) > 2 else "random 1e2..1e5 K", results=res)))
