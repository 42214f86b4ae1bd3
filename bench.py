#!/usr/bin/env python3
"""bench.py -- source*cell updates/s of the ASORA raytrace (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (config.workload): the reference's raytracing benchmark
(examples/paper_tests/raytracing_benchmark/run_test.py + parameters.yml) at the mesh BASELINE.json
names: 256^3 uniform grid (ndens 1e-3, xHII 2e-4, cell 3 Mpc/200), 1e4 random sources, R_max = 60
cells, black-body 1e5 K table (index 2.8, NumTau 20000).  One "step" = one do_all_sources call on
device-resident inputs (asora_do_all_sources_dev) over this rank's source shard, followed -- for
N > 1 -- by the RCCL all-reduce of phi_ion (the reference's Reduce+Bcast, pyc2ray/evolve.py:309-317).
The 1e4 sources are split over the ranks (evolve.py:175-186): total work is fixed => "strong".

value      = (sources * cells visited per source) * K / time, whole job, max time over ranks.
roofline   = algorithmic bytes (32 B per visit + 16 B per rate-producing visit) of
             one sweep-kernel launch / its HIP-event duration (events recorded by the library on
             the launch stream, averaged over the timed steps), against 8 TB/s.
cpu_baseline = the oracle timed on this node's host cores on a bounded sample (rank 0, N = 1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

# the host driver only supports dmabuf IPC (RCCL between processes); set before any HIP runtime loads
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MPC = 3.0856775814913673e24
HBM_PEAK_GBS = 8000.0


def make_inputs(a):
    from pyc2ray_amd.radiation import EV, H_PLANCK, BlackBodySource, make_tau_table
    N = a.N
    fmin, fmax = 13.598 * EV / H_PLANCK, 54.416 * EV / H_PLANCK
    tau, dlogtau = make_tau_table(-20.0, 4.0, a.numtau)
    bb = BlackBodySource(1e5, False, fmin, 2.8)
    thin, thick = bb.make_photo_table(tau, fmin, fmax, 1e48)
    hthin, hthick = bb.make_heat_table(tau, fmin, fmax, 1e48) if a.heat else (None, None)
    rng = np.random.default_rng(918)
    if a.density == "uniform":
        ndens = np.full((N, N, N), 1e-3)
        pos = rng.integers(0, N, (a.sources, 3)).astype(np.int32)
    else:  # log-normal density, sources drawn with probability ~ ndens^2
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from cases import lognormal_density
        ndens = lognormal_density(N, 7, smooth=2.0)
        p = (ndens ** 2).ravel()
        idx = np.random.default_rng(11).choice(p.size, size=a.sources, p=p / p.sum())
        pos = np.stack(np.unravel_index(idx, (N, N, N)), axis=1).astype(np.int32)
    # run_test.py:53-55: fact * uniform(1e9, 5e11) / 1e48 with fact ~ 4.09e40
    flux = 4.09e40 * rng.uniform(1e9, 5e11, a.sources) / 1e48
    xh = np.full((N, N, N), a.xh)
    return dict(ndens=ndens, xh=xh, pos=pos, flux=flux, thin=thin, thick=thick, hthin=hthin, hthick=hthick,
                dlogtau=dlogtau)


def cpu_baseline(a):
    cmd = [sys.executable, os.path.join(ROOT, "oracle", "cpu_baseline.py"), "--N", str(a.N), "--R", str(a.R),
           "--dr", repr(a.dr), "--xh", repr(a.xh), "--sources-per-core", str(a.cpu_sources_per_core),
           "--numtau", str(a.numtau)]
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, check=True).stdout
        r = json.loads(out.strip().splitlines()[-1])
    except Exception as e:  # the baseline is a reported number, never a reason to lose the GPU line
        return {"value": None, "unit": "source*cell updates/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
    return {"value": r["value"], "unit": "source*cell updates/s", "cores": r["cores"], "kind": r["kind"],
            "per_core": r["per_core"],
            "sample": (f"{r['sources']} of the {a.sources} sources, {a.N}^3, R={a.R:g}: cube sweep "
                       f"(2R+1)^3={r['cube_cells_per_source']} cells/source, {r['cores']} single-threaded "
                       f"processes x {a.cpu_sources_per_core} sources, raytrace time {r['seconds']:.1f} s")}


def _mark(msg):
    if os.environ.get("ASORA_BENCH_DEBUG"):
        import faulthandler
        faulthandler.dump_traceback_later(45, exit=False, file=sys.stderr)  # where is it, if it stops moving
        print(f"[bench rank {os.environ.get('RANK', '0')}] {msg}", file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--N", type=int, default=256)
    ap.add_argument("--sources", type=int, default=10000)
    ap.add_argument("--R", type=float, default=60.0)
    ap.add_argument("--dr", type=float, default=3.0 * MPC / 200)
    ap.add_argument("--xh", type=float, default=2e-4)
    ap.add_argument("--numtau", type=int, default=20000)
    ap.add_argument("--density", choices=["uniform", "clustered"], default="uniform")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong")
    ap.add_argument("--shard-order", choices=["curve", "list"], default="curve",
                    help="N > 1: shard the source list after sorting it along the library's Hilbert curve "
                         "(spatially compact shards) or as given")
    ap.add_argument("--heat", action="store_true", help="also accumulate the photo-heating rate (not the headline config)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--cpu-sources-per-core", type=int, default=48)  # ~15 s of CPU work per core
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    cpu = None
    if rank == 0 and world == 1 and a.gpus == 1 and not a.no_cpu:
        cpu = cpu_baseline(a)  # before this process touches the GPU

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback for the ASORA path)")
    # one GPU per rank; ranks wrap around only in single-GPU rehearsals of the N > 1 path
    # (ASORA_BENCH_BACKEND=gloo, since RCCL refuses two ranks on one device)
    ngpu = torch.cuda.device_count()
    dev_index = local_rank % ngpu
    torch.cuda.set_device(dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("ASORA_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)

    _mark("process group ready")
    from pyc2ray_amd import libasora as asora
    from pyc2ray_amd.sourceutils import shard_sources

    if a.scaling == "weak":
        a.sources = a.sources * world
    inp = make_inputs(a)
    N = a.N
    # single-GPU rehearsal of one rank's share of an N-GPU run: ASORA_BENCH_EMULATE="rank/world"
    emu = os.environ.get("ASORA_BENCH_EMULATE")
    shard_rank, shard_world = (int(x) for x in emu.split("/")) if emu else (rank, world)
    if shard_world > 1 and a.shard_order == "curve":
        # source-list preprocessing, once, outside the timed steps: contiguous shards of the
        # curve-sorted list are spatially compact, so the sources a GPU has in flight share more of their reads
        order = asora.source_order(np.ascontiguousarray(inp["pos"].ravel()), N, 2)
        inp["pos"], inp["flux"] = inp["pos"][order], inp["flux"][order]
    lo, hi = shard_sources(a.sources, shard_rank, shard_world)
    ns_local = hi - lo
    dev = torch.device("cuda", dev_index)
    ndens_t = torch.from_numpy(inp["ndens"].ravel()).to(dev)
    xh_t = torch.from_numpy(inp["xh"].ravel()).to(dev)
    phi_t = torch.zeros(N ** 3, dtype=torch.float64, device=dev)

    for kv in filter(None, os.environ.get("ASORA_OPTIONS", "").split(",")):  # tuning experiments: name=value,...
        k, v = kv.split("=")
        asora.set_option(k, int(v))
    asora.device_init(N, 64, dev_index, ngpu)
    asora.photo_table_to_device(np.ascontiguousarray(inp["thin"]), np.ascontiguousarray(inp["thick"]),
                                inp["thin"].shape[0])
    heat_t = None
    if a.heat:
        asora.heat_table_to_device(np.ascontiguousarray(inp["hthin"]), np.ascontiguousarray(inp["hthick"]),
                                   inp["hthin"].shape[0])
        heat_t = torch.zeros(N ** 3, dtype=torch.float64, device=dev)
    asora.density_from_device(ndens_t.data_ptr(), N)
    asora.source_data_to_device(np.ascontiguousarray(inp["pos"][lo:hi].ravel()),
                                np.ascontiguousarray(inp["flux"][lo:hi]), ns_local)
    num_tau = inp["thin"].shape[0]
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        asora.do_all_sources_dev(a.R, 0, 6.30e-18, a.dr, xh_t.data_ptr(), phi_t.data_ptr(),
                                 heat_t.data_ptr() if a.heat else 0, ns_local, N, -20.0, inp["dlogtau"], num_tau, stream)
        if world > 1:
            if dist.get_backend() == "nccl":
                dist.all_reduce(phi_t)  # RCCL sum over xGMI, in place on device memory
                if a.heat:
                    dist.all_reduce(heat_t)
            else:  # single-GPU rehearsal only: through the host
                h = phi_t.cpu()
                dist.all_reduce(h)
                phi_t.copy_(h)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    _mark("inputs on device")
    for _ in range(a.warmup):
        step()
    _mark("warm-up issued")
    fence()
    _mark("warm-up done")
    kern_ms = []
    t0 = time.perf_counter()
    for i in range(a.steps):
        step()
        _mark(f"step {i} issued")
        kern_ms.append(asora.last_sweep_ms())  # HIP events on the launch stream (waits for the sweep)
        _mark(f"step {i} timed")
    fence()
    dt = time.perf_counter() - t0
    _mark("timed steps done")
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    visits, _ = asora.visit_count(a.R, N)
    in_sphere = asora.geometry_census(a.R, a.dr, N)["owned_rate"]  # the oracle's floating-point sphere test
    rate_only = "skip_unobserved=1" in os.environ.get("ASORA_OPTIONS", "")
    if rate_only:  # experiment only: the cells outside the R-sphere are not swept, so they are not counted either
        visits = in_sphere
    total_visits = float(visits) * (ns_local if emu else a.sources)  # rehearsal: this GPU's share only
    value = total_visits * a.steps / dt
    k_ms = float(np.mean(kern_ms))
    algo_bytes = (32.0 * visits + (32.0 if a.heat else 16.0) * in_sphere) * ns_local  # +16 B per heating-rate update
    achieved = algo_bytes / (k_ms * 1e-3) / 1e9
    info = asora.last_launch_info()
    traffic = None
    tf = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(tf) and not a.heat and not rate_only:  # the counters were collected on the default path
        try:
            tj = json.load(open(tf))
            key = f"{a.density}_N{N}_R{a.R:g}_ns{ns_local}"
            traffic = tj.get(key, {}).get("bytes_per_launch")
        except Exception:
            traffic = None
    checksum = float(phi_t.sum().item())
    asora.device_close()
    # second denominator: what a plain device-to-device copy achieves on this box
    copy_gbs = None
    if rank == 0:
        src_t = torch.empty(1 << 27, dtype=torch.float64, device=dev)  # 1 GiB
        dst_t = torch.empty_like(src_t)
        dst_t.copy_(src_t)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            dst_t.copy_(src_t)
        e1.record()
        torch.cuda.synchronize()
        copy_gbs = 5 * 2.0 * src_t.numel() * 8 / (e0.elapsed_time(e1) * 1e-3) / 1e9
        del src_t, dst_t

    if rank == 0:
        line = {
            "metric": "source*cell updates/s, ASORA raytrace", "value": value, "unit": "source*cell updates/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": a.scaling, "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"raytracing_benchmark: {N}^3 {a.density} grid, {a.sources} sources, "
                                   f"R_max={a.R:g} cells, cell {a.dr:.4e} cm, xHII {a.xh:g}, NumTau {a.numtau}"
                                   + (" [skip_unobserved: rate-producing cells only]" if rate_only else "")
                                   + (" [+ photo-heating rate]" if a.heat else "")
                                   + (f" [rehearsal of rank {emu}]" if emu else ""),
                       "visits_per_source": visits, "rate_visits_per_source": in_sphere,
                       "sources_per_gpu": ns_local, "parallelism": f"source-shard x{world} + all-reduce(phi_ion)",
                       "ns_per_source_per_sphere_cell": dt / a.steps / a.sources / (4.0 / 3.0 * np.pi * a.R ** 3) * 1e9,
                       "phi_checksum": checksum},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "measured_copy_GBs": copy_gbs, "frac_of_measured_copy": achieved / copy_gbs if copy_gbs else None, "kernel": "asora::sweep_tab_kernel" if info["table_driven"] else "asora::sweep_kernel",
                         "kernel_ms": k_ms, "algorithmic_bytes_per_launch": algo_bytes,
                         "launch": info},
        }
        if cpu is not None:
            line["cpu_baseline"] = cpu
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
