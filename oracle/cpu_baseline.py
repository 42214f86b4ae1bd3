"""CPU baseline leg of bench.py -- times the oracle (the compiled reference Fortran when
oracle/_ref/libc2ray_ref.so is present, else the C restatement) on a BOUNDED sample of the bench
workload, on the host cores of the machine it runs on.  Runs as its own process, before bench.py
touches the GPU, and prints one JSON object.  TEST/MEASUREMENT INFRASTRUCTURE ONLY.

The CPU algorithm sweeps the cube src +- min(max_subbox, N/2) (raytracing.f90:174-175), so for
R = 60 it is run with max_subbox = subboxsize = R and loss_fraction = 0: (2R+1)^3 cube cells per
source against the GPU's octahedron cells -- both are reported in the same unit, (source, cell)
evaluations per second of each algorithm's own visit set.
"""
from __future__ import annotations

import argparse
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _worker(job):
    kind, N, R, dr, sig, xh0, nd0, seed, pos, flux, numtau = job
    from oracle.oracle import PortOracle, RefOracle
    from pyc2ray_amd.radiation import BlackBodySource, make_tau_table
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import cases
    orc = RefOracle() if kind == "reference" else PortOracle()
    tau, dlogtau = make_tau_table(-20.0, 4.0, numtau)
    bb = BlackBodySource(1e5, False, cases.FREQ_MIN, 2.8)
    thin, thick = bb.make_photo_table(tau, cases.FREQ_MIN, cases.FREQ_MAX, 1e48)
    nd = np.full((N, N, N), nd0)
    xh = np.full((N, N, N), xh0)
    box = int(min(R, N // 2))
    t0 = time.perf_counter()
    orc.do_all_sources(flux, pos, nd, xh, sig, dr, thin, thick, -20.0, dlogtau, float(R), max_subbox=box,
                       subboxsize=box, loss_fraction=0.0, flux_mode=0)  # one call, like the reference benchmark
    return time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--N", type=int, default=256)
    ap.add_argument("--R", type=float, default=60.0)
    ap.add_argument("--dr", type=float, required=True)
    ap.add_argument("--sig", type=float, default=6.30e-18)
    ap.add_argument("--xh", type=float, default=2e-4)
    ap.add_argument("--ndens", type=float, default=1e-3)
    ap.add_argument("--cores", type=int, default=0)
    ap.add_argument("--sources-per-core", type=int, default=8)
    ap.add_argument("--numtau", type=int, default=20000)
    a = ap.parse_args()
    kind = "reference" if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libc2ray_ref.so")) else "port"
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = a.cores or min(16, avail)
    rng = np.random.default_rng(918)
    ns = cores * a.sources_per_core
    pos = rng.integers(0, a.N, (ns, 3)).astype(np.int32)
    flux = rng.uniform(40.0, 2.0e4, ns)
    jobs = [(kind, a.N, a.R, a.dr, a.sig, a.xh, a.ndens, i, pos[i::cores], flux[i::cores], a.numtau)
            for i in range(cores)]
    t0 = time.perf_counter()
    with mp.get_context("spawn").Pool(cores) as pool:
        per = pool.map(_worker, jobs)
    wall = time.perf_counter() - t0
    box = int(min(a.R, a.N // 2))
    side = min(2 * box + 1, a.N)
    cube = side ** 3
    # throughput of the node's cores working in parallel: all sources / slowest worker's raytrace time
    # (table generation and process start-up excluded)
    t_rt = max(per)
    print(json.dumps(dict(kind=kind, cores=cores, sources=ns, cube_cells_per_source=cube, seconds=t_rt,
                          wall_seconds=wall, value=ns * cube / t_rt,
                          per_core=float(np.mean([a.sources_per_core * cube / t for t in per])))))


if __name__ == "__main__":
    main()
