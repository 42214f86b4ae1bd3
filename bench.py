#!/usr/bin/env python3
"""bench.py -- source*cell updates/s of the ASORA raytrace (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

With --gpus N > 1 and no launcher environment (WORLD_SIZE unset) this process only LAUNCHES: it starts N fresh
ranks through torch.distributed.run as a child (before anything here touches the GPU), relays their output and
exit code.  A rank whose WORLD_SIZE differs from --gpus exits non-zero.

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
secondary  = (N = 1, default workload only) the regimes the headline does not show, each a few device-resident
             steps timed by the same HIP events: clustered sources, the heating output, the reference's own
             published point (N = 200, R = 10), BASELINE configs [0], [3], [4] on one GPU, small and large radii.
             Same `frac` definition as `roofline`; `value` / `roofline` are untouched by it.
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


def _clustered_inputs(N, nsrc, base256=None):
    """Log-normal density and sources drawn with probability ~ ndens^2.  512^3: the periodic
    256^3 field tiled twice per axis (an FFT of 512^3 costs more than all the timed steps together)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from cases import lognormal_density
    if N == 512:
        f = base256 if base256 is not None else lognormal_density(256, 7, smooth=2.0)
        ndens = np.ascontiguousarray(np.tile(f, (2, 2, 2)))
    elif N == 256 and base256 is not None:
        ndens = base256
    else:
        ndens = lognormal_density(N, 7, smooth=2.0)
    p = (ndens ** 2).ravel()
    cdf = np.cumsum(p)
    idx = np.searchsorted(cdf, np.random.default_rng(11).random(nsrc) * cdf[-1])
    pos = np.stack(np.unravel_index(np.minimum(idx, p.size - 1), (N, N, N)), axis=1).astype(np.int32)
    return ndens, pos


# (label, N, sources, R, density, heat, cdh) -- the headline's table, cell size and xHII throughout
SECONDARY = [
    ("256^3 clustered grid, 1e4 clustered sources, R=60", 256, 10000, 60.0, "clustered", False, False),
    ("headline workload + photo-heating output", 256, 10000, 60.0, "uniform", True, False),
    ("headline workload + coldensh_out exported (what the host API does)", 256, 10000, 60.0, "uniform", False, True),
    ("reference's published point: N=200, R=10, 1e5 sources", 200, 100000, 10.0, "uniform", False, False),
    ("BASELINE configs[3] share: 250^3 clustered, 12500 sources, R=58.6", 250, 12500, 58.6, "clustered", False, False),
    ("BASELINE configs[4] share: 512^3 clustered, 500 sources, R=150", 512, 500, 150.0, "clustered", False, False),
    ("BASELINE configs[0]/[2]: 128^3 full box, 1 source", 128, 1, 150.0, "uniform", False, False),
    ("256^3 uniform, 1e4 sources, R=3", 256, 10000, 3.0, "uniform", False, False),
    ("256^3 uniform, 1e4 sources, R=5", 256, 10000, 5.0, "uniform", False, False),
    ("256^3 uniform, 1e4 sources, R=10", 256, 10000, 10.0, "uniform", False, False),
    ("256^3 uniform, 1e4 sources, R=30", 256, 10000, 30.0, "uniform", False, False),
    ("256^3 uniform, 1000 sources, R=10 (short call: fixed per-call kernels)", 256, 1000, 10.0, "uniform", False, False),
    ("256^3 uniform, 1000 sources, R=160", 256, 1000, 160.0, "uniform", False, False),
]


def secondary_runs(a, asora, torch, dev, inp, steps=3, warmup=1):
    """The regimes beside the headline.  Every entry: device-resident inputs, `warmup` + `steps`
    calls of asora_do_all_sources_dev, kernel time from the library's HIP events on the launch stream, wall time per
    call from the host clock around synchronised steps."""
    out = []
    stream = torch.cuda.current_stream().cuda_stream
    num_tau = inp["thin"].shape[0]
    base256 = None
    hthin = hthick = None
    for label, N, nsrc, R, density, heat, cdh in SECONDARY:
        try:
            rng = np.random.default_rng(918)
            if density == "uniform":
                ndens = np.full((N, N, N), 1e-3)
                pos = rng.integers(0, N, (nsrc, 3)).astype(np.int32)
            else:
                if base256 is None and N in (256, 512):
                    base256, _ = _clustered_inputs(256, 1)
                ndens, pos = _clustered_inputs(N, nsrc, base256)
            flux = 4.09e40 * rng.uniform(1e9, 5e11, nsrc) / 1e48
            asora.device_init(N, 64, dev.index, torch.cuda.device_count())
            asora.photo_table_to_device(np.ascontiguousarray(inp["thin"]), np.ascontiguousarray(inp["thick"]), num_tau)
            if heat:
                if hthin is None:
                    from pyc2ray_amd.radiation import EV, H_PLANCK, BlackBodySource, make_tau_table
                    fmin, fmax = 13.598 * EV / H_PLANCK, 54.416 * EV / H_PLANCK
                    tau, _ = make_tau_table(-20.0, 4.0, a.numtau)
                    hthin, hthick = BlackBodySource(1e5, False, fmin, 2.8).make_heat_table(tau, fmin, fmax, 1e48)
                asora.heat_table_to_device(np.ascontiguousarray(hthin), np.ascontiguousarray(hthick), num_tau)
            ndens_t = torch.from_numpy(np.ascontiguousarray(ndens).ravel()).to(dev)
            xh_t = torch.full((N ** 3,), a.xh, dtype=torch.float64, device=dev)
            phi_t = torch.zeros(N ** 3, dtype=torch.float64, device=dev)
            heat_t = torch.zeros(N ** 3, dtype=torch.float64, device=dev) if heat else None
            cdh_t = torch.zeros(N ** 3, dtype=torch.float64, device=dev) if cdh else None
            asora.density_from_device(ndens_t.data_ptr(), N)
            asora.source_data_to_device(np.ascontiguousarray(pos.ravel()), np.ascontiguousarray(flux), nsrc)

            def call():
                asora.do_all_sources_dev(R, cdh_t.data_ptr() if cdh else 0, 6.30e-18, a.dr, xh_t.data_ptr(),
                                         phi_t.data_ptr(), heat_t.data_ptr() if heat else 0, nsrc, N, -20.0,
                                         inp["dlogtau"], num_tau, stream)
            for _ in range(warmup):
                call()
            torch.cuda.synchronize()
            k_ms, t0 = [], time.perf_counter()
            for _ in range(steps):
                call()
                k_ms.append(asora.last_sweep_ms())
            torch.cuda.synchronize()
            wall_ms = (time.perf_counter() - t0) / steps * 1e3
            k = float(np.mean(k_ms))
            visits, _ = asora.visit_count(R, N)
            in_sphere = asora.geometry_census(R, a.dr, N)["owned_rate"]
            algo = (32.0 * visits + (32.0 if heat else 16.0) * in_sphere) * nsrc
            info = asora.last_launch_info()
            e = {"workload": label, "N": N, "sources": nsrc, "R": R, "kernel_ms": k, "ms_per_step": wall_ms,
                 "updates_per_s": float(visits) * nsrc / (k * 1e-3), "frac": algo / (k * 1e-3) / 1e9 / HBM_PEAK_GBS,
                 "visits_per_source": visits, "rate_visits_per_source": in_sphere,
                 "launch": f"{info['block']} thr, {info['staging']}, {info['sectors_per_workgroup']} sector(s) x "
                           f"{info['sources_per_workgroup']} source(s), {info['lds_bytes']} B LDS"
                           + ("" if info["table_driven"] else " [table-free fallback kernel]")}
            if N == 200:  # the reference's own unit (docs/fig/scaling.jpg): published 1.466 ns (A100), 1.081 ns (GH200)
                e["ns_per_source_per_sphere_cell"] = wall_ms * 1e-3 / nsrc / (4.0 / 3.0 * np.pi * R ** 3) * 1e9
            out.append(e)
            del ndens_t, xh_t, phi_t, heat_t, cdh_t
        except Exception as ex:  # a secondary line never costs the headline
            out.append({"workload": label, "error": f"{type(ex).__name__}: {ex}"})
        finally:
            try:
                asora.device_close()
            except Exception:
                pass
    return out


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
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="form the process group, count the ranks, print {n_gpus, ranks_seen, backend} and exit "
                         "(no GPU work: checks the launcher / rendezvous path of a node)")
    ap.add_argument("--no-host-api", action="store_true", help="skip the PCIe-inclusive host-pointer measurement")
    ap.add_argument("--secondary", choices=["auto", "on", "off"], default="auto",
                    help="the `secondary` list of the JSON line (auto: with the default workload on one GPU)")
    ap.add_argument("--cpu-sources-per-core", type=int, default=48)  # ~15 s of CPU work per core
    a = ap.parse_args()

    if a.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # launcher: N fresh ranks, one per GPU; this process never initialises the GPU (no exec of a GPU process)
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
        raise SystemExit(subprocess.run(cmd).returncode)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if a.rendezvous_only:
        import torch
        import torch.distributed as dist
        backend = os.environ.get("ASORA_BENCH_BACKEND", "gloo")
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group(backend)
        seen = torch.ones(1, dtype=torch.int64)
        if world > 1:
            dist.all_reduce(seen)
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"n_gpus": world, "ranks_seen": int(seen.item()), "backend": backend if world > 1 else None,
                              "rendezvous_only": True}), flush=True)
        return

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
    backend = None
    # ASORA_BENCH_FORCE_DIST=1: form the process group and run the collective even with ONE rank -- the only way to
    # put the RCCL all-reduce and its stream ordering against the sweep under test on a single-GPU box
    dist_on = world > 1 or os.environ.get("ASORA_BENCH_FORCE_DIST") == "1"
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1]))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
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

    ar_events = []  # (start, stop) events around the collective of each step, on the launch stream

    def step():
        asora.do_all_sources_dev(a.R, 0, 6.30e-18, a.dr, xh_t.data_ptr(), phi_t.data_ptr(),
                                 heat_t.data_ptr() if a.heat else 0, ns_local, N, -20.0, inp["dlogtau"], num_tau, stream)
        if dist_on:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            if backend == "nccl":
                # RCCL sum over xGMI, in place on device memory.  The sweep was launched on torch's current stream
                # (`stream` above) and ProcessGroupNCCL orders its own stream behind the current one on entry and the
                # current one behind the collective on return: sweep -> all-reduce -> next step's kernels.
                dist.all_reduce(phi_t)
                if a.heat:
                    dist.all_reduce(heat_t)
            else:  # single-GPU rehearsal only: through the host
                h = phi_t.cpu()
                dist.all_reduce(h)
                phi_t.copy_(h)
            e1.record()
            ar_events.append((e0, e1))

    def fence():
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    _mark("inputs on device")
    for _ in range(a.warmup):
        step()
    _mark("warm-up issued")
    fence()
    _mark("warm-up done")
    kern_ms = []
    ar_events.clear()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step()
        _mark(f"step {i} issued")
        kern_ms.append(asora.last_sweep_ms())  # HIP events on the launch stream (waits for the sweep)
        _mark(f"step {i} timed")
    fence()
    dt = time.perf_counter() - t0
    _mark("timed steps done")
    # what every rank saw: its sweep-kernel time, its wait + transfer in the collective, and a head count
    k_ms = float(np.mean(kern_ms))
    ar_ms = float(np.mean([e0.elapsed_time(e1) for e0, e1 in ar_events])) if ar_events else 0.0
    per_rank = torch.tensor([dt, k_ms, ar_ms, 1.0, float(ns_local)], dtype=torch.float64, device=dev)
    if dist_on:
        gathered = [torch.empty_like(per_rank) for _ in range(world)]
        if backend != "nccl":
            gathered = [g.cpu() for g in gathered]
            dist.all_gather(gathered, per_rank.cpu())
        else:
            dist.all_gather(gathered, per_rank)
        allr = torch.stack([g.cpu() for g in gathered]).numpy()
    else:
        allr = per_rank.cpu().numpy()[None, :]
    dt = float(allr[:, 0].max())
    ranks_seen = int(round(allr[:, 3].sum()))

    visits, _ = asora.visit_count(a.R, N)
    in_sphere = asora.geometry_census(a.R, a.dr, N)["owned_rate"]  # the oracle's floating-point sphere test
    rate_only = "skip_unobserved=1" in os.environ.get("ASORA_OPTIONS", "")
    if rate_only:  # experiment only: the cells outside the R-sphere are not swept, so they are not counted either
        visits = in_sphere
    total_visits = float(visits) * (ns_local if emu else a.sources)  # rehearsal: this GPU's share only
    value = total_visits * a.steps / dt
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
    # PCIe-inclusive rate of the host-pointer entry point the reference's callers use (asora_do_all_sources: 1 x N^3
    # up, 1-2 x N^3 down per call, src/asora/raytracing.cu:130,163-166) -- reported beside the headline, never as it
    host_api = None
    if rank == 0 and world == 1 and not emu and not a.no_host_api and not a.heat:
        asora.density_to_device(np.ascontiguousarray(inp["ndens"].ravel()), N)
        xh_h, phi_h, cdh_h = np.ascontiguousarray(inp["xh"].ravel()), np.zeros(N ** 3), np.zeros(N ** 3)
        hs = []
        for _ in range(3):
            th = time.perf_counter()
            asora.do_all_sources(a.R, cdh_h, 6.30e-18, a.dr, None, xh_h, phi_h, ns_local, N, -20.0, inp["dlogtau"], num_tau)
            hs.append(time.perf_counter() - th)
        host_api = {"ms_per_call": min(hs[1:]) * 1e3, "value": total_visits / min(hs[1:]),
                    "first_call_ms": hs[0] * 1e3,
                    # the reference's own normalisation (docs/fig/scaling.jpg): wall time of this very call per source
                    # and per cell of the R-sphere; published asymptotes at N = 200, R = 10: 1.466 ns (A100), 1.081 ns (GH200)
                    "ns_per_source_per_sphere_cell": min(hs[1:]) / a.sources / (4.0 / 3.0 * np.pi * a.R ** 3) * 1e9,
                    "what": "asora_do_all_sources on host pointers: H2D xh_av + sweep + D2H phi_ion and coldensh_out"}
    asora.device_close()
    secondary = None
    default_workload = (N == 256 and a.sources == 10000 and a.R == 60.0 and a.density == "uniform" and not a.heat
                        and not os.environ.get("ASORA_OPTIONS"))
    if rank == 0 and world == 1 and not emu and not dist_on and (a.secondary == "on" or (a.secondary == "auto" and default_workload)):
        del ndens_t, xh_t, phi_t
        _mark("secondary runs")
        secondary = secondary_runs(a, asora, torch, dev, inp)
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
            "n_gpus": world, "ranks_seen": ranks_seen, "backend": backend, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
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
            "per_rank": {"kernel_ms": [float(v) for v in allr[:, 1]], "allreduce_ms": [float(v) for v in allr[:, 2]],
                         "step_ms": [float(v) / a.steps * 1e3 for v in allr[:, 0]],
                         "sources": [int(v) for v in allr[:, 4]]},
            # `achieved` / `frac` are ALGORITHMIC bytes (32 B per visit + 16 B per rate visit) over the
            # kernel time against the HBM peak -- the reporting convention of this tier, not measured HBM traffic: the
            # column density lives in LDS, so the fabric moves about half of that (`traffic`) and the kernel's actual
            # limiter is VALU issue (profiles/r*_pmc_summary.json)
            "roofline": {"bound": "hbm", "limiter": "valu-issue (nominal bound: hbm)", "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "frac_is": "algorithmic bytes / kernel time / peak",
                         "traffic": traffic,
                         "traffic_source": ("profiles/hbm_traffic.json (rocprofv3 --pmc passes on this workload, "
                                            "not this run)") if traffic is not None else None,
                         "traffic_frac_of_peak": (traffic / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                         "measured_copy_GBs": copy_gbs, "frac_of_measured_copy": achieved / copy_gbs if copy_gbs else None, "kernel": "asora::sweep_tab_kernel" if info["table_driven"] else "asora::sweep_kernel",
                         "kernel_ms": k_ms, "algorithmic_bytes_per_launch": algo_bytes,
                         "launch": info},
        }
        if secondary is not None:
            line["secondary"] = secondary
        if host_api is not None:
            line["host_api"] = host_api
        if cpu is not None:
            line["cpu_baseline"] = cpu
        print(json.dumps(line), flush=True)
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
