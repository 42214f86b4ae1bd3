"""The N > 1 paths on the GPU -- evolve3D and bench.py: two ranks (one GPU here, so the all-reduce goes through gloo and
the host -- ASORA_BENCH_BACKEND=gloo; RCCL refuses two ranks on one device) shard the source list and
must reproduce the single-rank rate grid."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARGS = ["--N", "64", "--sources", "481", "--R", "20", "--no-cpu", "--steps", "1", "--warmup", "1"]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _last_json(out):
    return json.loads([l for l in out.strip().splitlines() if l.startswith("{")][-1])


@pytest.mark.gpu
def test_two_rank_bench_reproduces_single_rank_checksum():
    env = dict(os.environ, ASORA_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *ARGS], capture_output=True, text=True,
                         timeout=300, cwd=ROOT, env=env)
    assert one.returncode == 0, one.stderr[-2000:]
    # the driver's plain command form: bench.py itself starts the two ranks
    two = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", *ARGS], capture_output=True,
                         text=True, timeout=300, cwd=ROOT, env={k: v for k, v in env.items() if k not in
                                                                 ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
    assert two.returncode == 0, two.stderr[-2000:]
    a, b = _last_json(one.stdout), _last_json(two.stdout)
    assert b["n_gpus"] == 2 and b["ranks_seen"] == 2 and b["backend"] == "gloo"
    assert b["per_rank"]["sources"] == [240, 241] and len(b["per_rank"]["kernel_ms"]) == 2
    assert all(v > 0 for v in b["per_rank"]["kernel_ms"] + b["per_rank"]["allreduce_ms"])
    assert a["n_gpus"] == 1 and a["ranks_seen"] == 1 and a["per_rank"]["allreduce_ms"] == [0.0]
    # and under an external launcher (the driver's other form)
    ext = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                          os.path.join(ROOT, "bench.py"), "--gpus", "2", *ARGS], capture_output=True, text=True,
                         timeout=300, cwd=ROOT, env=env)
    assert ext.returncode == 0, ext.stderr[-2000:]
    assert _last_json(ext.stdout)["ranks_seen"] == 2
    ca, cb = a["config"]["phi_checksum"], b["config"]["phi_checksum"]
    assert ca > 0 and abs(ca - cb) <= 1e-12 * ca, (ca, cb)  # sum order differs, nothing else


@pytest.mark.gpu
def test_rccl_allreduce_follows_the_sweep_on_its_stream():
    """RCCL itself on the bench path (a one-GPU box cannot host two RCCL ranks): ASORA_BENCH_FORCE_DIST=1 forms a
    one-rank "nccl" process group, so every step runs sweep -> dist.all_reduce(phi_ion) through RCCL on torch's
    current stream -- the stream the library launched the sweep on.  A missing ordering between the two would
    reduce a half-written grid: the checksum must equal the plain run's, over several steps."""
    args = ["--N", "64", "--sources", "481", "--R", "20", "--no-cpu", "--no-host-api", "--steps", "4", "--warmup", "1"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    plain = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True,
                           timeout=300, cwd=ROOT, env=env)
    assert plain.returncode == 0, plain.stderr[-2000:]
    rccl = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True,
                          timeout=300, cwd=ROOT, env=dict(env, ASORA_BENCH_FORCE_DIST="1"))
    assert rccl.returncode == 0, rccl.stderr[-2000:]
    a, b = _last_json(plain.stdout), _last_json(rccl.stdout)
    assert a["backend"] is None and b["backend"] == "nccl" and b["ranks_seen"] == 1 and b["n_gpus"] == 1
    assert b["per_rank"]["allreduce_ms"][0] > 0.0
    ca, cb = a["config"]["phi_checksum"], b["config"]["phi_checksum"]
    assert ca > 0 and abs(ca - cb) <= 1e-12 * ca, (ca, cb)


def _run_evolve_ranks(world, out_dir, case_name):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                        os.path.join(ROOT, "tests", "evolve_rank_worker.py"), str(out_dir), case_name],
                       capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    import numpy as np
    return [np.load(os.path.join(out_dir, f"evolve_{k}.npz")) for k in range(world)]


@pytest.mark.gpu
def test_evolve3D_two_ranks_equal_the_single_rank_golden(tmp_path):
    """pyc2ray_amd.evolve.evolve3D with world_size 2 (reference: pyc2ray/evolve.py:147-154, 173-187, 309-317,
    386-398): the three sources of the evolve32 case sharded 1 + 2 over two fresh processes on the one GPU.  Both
    ranks must return identical grids; iteration count and per-iteration conv_flag must be those of the
    single-rank golden made from the compiled reference (the criterion comes from the GLOBAL source count), xHII
    within north_star's 1e-8."""
    import numpy as np
    g = np.load(os.path.join(ROOT, "tests", "golden", "evolve32.npz"))
    r0, r1 = _run_evolve_ranks(2, tmp_path, "evolve32")
    assert np.array_equal(r0["xh_new"], r1["xh_new"]) and np.array_equal(r0["phi_ion"], r1["phi_ion"])
    assert float(r0["conv_criterion"]) == float(r1["conv_criterion"]) == (3 - 1) / 3
    assert int(r0["niter"]) == int(r1["niter"]) == int(g["niter"])
    assert list(r0["conv_flags"]) == list(r1["conv_flags"]) == list(g["conv_flags"])
    err = np.max(np.abs(r0["xh_new"] - g["xh_new"]))
    print(f"[evolve32, 2 ranks] niter {int(r0['niter'])}  max|d xHII| {err:.2e}")
    assert err <= 1e-8
    m = g["phi_ion"] > 0
    assert np.max(np.abs(r0["phi_ion"][m] / g["phi_ion"][m] - 1.0)) <= 1e-8


@pytest.mark.gpu
def test_evolve3D_four_ranks_ten_sources(tmp_path):
    """10 sources on 4 ranks (2, 2, 2, 4): the criterion is the reference's 3.0 on every rank and the loop ends
    where the single-rank run ends."""
    import numpy as np
    (tmp_path / "one").mkdir()
    one = _run_evolve_ranks(1, tmp_path / "one", "ten_sources16")
    rs = _run_evolve_ranks(4, tmp_path, "ten_sources16")
    for r in rs:
        assert float(r["conv_criterion"]) == 3.0
        assert int(r["niter"]) == int(one[0]["niter"]) and list(r["conv_flags"]) == list(one[0]["conv_flags"])
        assert np.array_equal(r["xh_new"], rs[0]["xh_new"])
        assert np.max(np.abs(r["xh_new"] - one[0]["xh_new"])) <= 1e-8
    assert 1 <= int(one[0]["conv_flags"][-1]) < 3
