"""world_size-2 gloo test of the multi-GPU host logic (shard + all-reduce), with the CPU oracle
standing in for the per-rank raytrace (the HIP path itself is covered by -m gpu tests)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank_main(rank, world, port, out_dir):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import cases
    from oracle.oracle import PortOracle
    from pyc2ray_amd.distributed import raytrace_sharded
    orc = PortOracle()
    c = cases.case_multi16_unequal()
    N = c["N"]
    seen = {}

    def local(pos_flat, flux, ns, phi_out, heat_out):
        seen["ns"] = ns
        if ns == 0:
            phi_out.zero_()
            return
        o = cases.run_oracle(orc, dict(c, pos=pos_flat.reshape(-1, 3), flux=flux), heat=False)
        phi_out.copy_(torch.from_numpy(o["phi_ion"].ravel()))

    phi = torch.empty(N ** 3, dtype=torch.float64)
    raytrace_sharded(local, c["pos"].T, c["flux"], phi)
    np.save(os.path.join(out_dir, f"phi_{rank}.npy"), phi.numpy())
    np.save(os.path.join(out_dir, f"ns_{rank}.npy"), np.array(seen["ns"]))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_shard_and_allreduce_world2(tmp_path, port_oracle):
    import cases
    world = 2
    mp.spawn(_rank_main, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    c = cases.case_multi16_unequal()
    ref = cases.run_oracle(port_oracle, c, heat=False)["phi_ion"].ravel()
    p0, p1 = (np.load(tmp_path / f"phi_{r}.npy") for r in range(world))
    assert np.array_equal(p0, p1)                       # every rank holds the full sum (Reduce+Bcast semantics)
    assert int(np.load(tmp_path / "ns_0.npy")) + int(np.load(tmp_path / "ns_1.npy")) == 16
    m = ref != 0
    assert np.max(np.abs(p0[m] - ref[m]) / ref[m]) < 1e-13   # only the summation order differs
    assert np.array_equal(p0 != 0, m)


def _bench(*args, env=None):
    import subprocess
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True,
                          timeout=300, cwd=ROOT, env=e)


@pytest.mark.timeout(300)
def test_bench_gpus_n_launches_n_ranks():
    """`python bench.py --gpus 2` with no launcher environment must START two ranks (torch.distributed.run as a
    child, before anything touches a GPU) and count them; --rendezvous-only stops after the head count, so this
    runs on CPU with gloo."""
    import json
    r = _bench("--gpus", "2", "--rendezvous-only", env={"ASORA_BENCH_BACKEND": "gloo"})
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["backend"] == "gloo"


def test_bench_refuses_a_world_that_differs_from_gpus():
    r = _bench("--gpus", "4", "--rendezvous-only", env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr
    r = _bench("--gpus", "1", "--rendezvous-only")
    assert r.returncode == 0 and '"ranks_seen": 1' in r.stdout
