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


def _evolve_rank_main(rank, world, port, out_dir, case_name):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from evolve_host import evolve3D_host_sharded
    from oracle.oracle import PortOracle
    r = evolve3D_host_sharded(PortOracle(), _EVOLVE_CASES[case_name]())
    np.savez(os.path.join(out_dir, f"evolve_{rank}.npz"), xh_new=r["xh_new"], phi_ion=r["phi_ion"], niter=r["niter"],
             conv_flags=np.array(r["conv_flags"]), n_local=r["n_local"], conv_criterion=r["conv_criterion"])
    dist.destroy_process_group()


def _evolve_case10():
    """10 sources in a 16^3 box with convergence_fraction 1e-3: int(1e-3 * 4096) = 4 > (10 - 1) / 3 = 3.0, so the
    source-count term of the criterion binds (pyc2ray/evolve.py:154).  Formed per shard it would be 0.33 (two
    sources on ranks 0-2) and 1.0 (four on rank 3)."""
    import cases
    c = cases.evolve_case(N=16, seed=9)
    rng = np.random.default_rng(77)
    c["pos"] = rng.integers(0, 16, (10, 3)).astype(np.int32)
    c["flux"] = rng.uniform(0.05, 2.0, 10)
    c["convergence_fraction"] = 1e-3
    return c


def _evolve_case32():
    import cases
    return cases.evolve_case()


_EVOLVE_CASES = {"evolve32": _evolve_case32, "ten_sources16": _evolve_case10}


@pytest.mark.timeout(600)
def test_evolve_step_world2_equals_single_rank_golden(tmp_path, golden_dir):
    """The evolve32 step with its three sources sharded 1 + 2 over two gloo ranks, every device operation done by
    the port oracle, the host logic being the product's (_sharded_step): both ranks return the same grids, and
    iteration count, per-iteration conv_flag and xHII are those of the single-rank golden (made from the compiled
    reference).  With the criterion formed from the shard sizes (0 on rank 0, whose decision is broadcast) the
    loop would run on until the relative-change test ends it."""
    world = 2
    mp.spawn(_evolve_rank_main, args=(world, _free_port(), str(tmp_path), "evolve32"), nprocs=world, join=True)
    g = np.load(os.path.join(golden_dir, "evolve32.npz"))
    r0, r1 = (np.load(tmp_path / f"evolve_{r}.npz") for r in range(world))
    assert [int(r0["n_local"]), int(r1["n_local"])] == [1, 2]
    assert np.array_equal(r0["xh_new"], r1["xh_new"]) and np.array_equal(r0["phi_ion"], r1["phi_ion"])
    assert float(r0["conv_criterion"]) == float(r1["conv_criterion"]) == (3 - 1) / 3
    assert int(r0["niter"]) == int(r1["niter"]) == int(g["niter"])
    assert list(r0["conv_flags"]) == list(g["conv_flags"])
    assert np.max(np.abs(r0["xh_new"] - g["xh_new"])) <= 1e-12
    m = g["phi_ion"] > 0
    assert np.max(np.abs(r0["phi_ion"][m] / g["phi_ion"][m] - 1.0)) <= 1e-10


@pytest.mark.timeout(600)
def test_evolve_step_world4_criterion_is_global(tmp_path, port_oracle):
    """10 sources on 4 ranks (shards 2, 2, 2, 4): every rank forms the reference's criterion 3.0 from the global
    list and the loop iterates exactly like the single-rank run."""
    from evolve_host import evolve3D_host_sharded
    world = 4
    mp.spawn(_evolve_rank_main, args=(world, _free_port(), str(tmp_path), "ten_sources16"), nprocs=world, join=True)
    one = evolve3D_host_sharded(port_oracle, _evolve_case10())      # no process group here: single rank
    assert one["conv_criterion"] == 3.0 and one["n_local"] == 10
    rs = [np.load(tmp_path / f"evolve_{r}.npz") for r in range(world)]
    assert [int(r["n_local"]) for r in rs] == [2, 2, 2, 4]
    for r in rs:
        assert float(r["conv_criterion"]) == 3.0
        assert int(r["niter"]) == one["niter"] and list(r["conv_flags"]) == one["conv_flags"]
        assert np.array_equal(r["xh_new"], rs[0]["xh_new"])
        assert np.max(np.abs(r["xh_new"] - one["xh_new"])) <= 1e-12
    # the case does discriminate: the loop ends on a conv_flag between the per-shard criteria (0.33 on rank 0,
    # whose decision is broadcast) and the global one
    assert 1 <= one["conv_flags"][-1] < 3


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
