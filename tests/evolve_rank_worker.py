"""One rank of a multi-rank evolve3D run on the GPU (started by tests/test_gpu_multirank.py through
torch.distributed.run).  All ranks share the box's one GPU, so the process group is gloo (RCCL refuses two
ranks on one device); everything else is the product path: pyc2ray_amd.evolve.evolve3D on the HIP library."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)


def main(out_dir, case_name):
    import torch.distributed as dist

    import cases
    from pyc2ray_amd import asora_core
    from pyc2ray_amd.evolve import evolve3D
    from test_distributed_cpu import _EVOLVE_CASES
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    c = _EVOLVE_CASES[case_name]()
    N = c["N"]
    asora_core.device_init(N, 8, rank, 1)
    try:
        asora_core.photo_table_to_device(c["thin"], c["thick"])
        hist = []
        xh_new, phi, cdh = evolve3D(c["dt"], c["dr"], c["flux"], c["pos"].T, True, 0, 0, 0.0, world > 1, None, rank,
                                    world, c["temp"], c["ndens"], c["xh"], c["clump"], c["thin"], c["thick"],
                                    c["minlogtau"], c["dlogtau"], c["R"], c["convergence_fraction"], c["sig"],
                                    *[c[k] for k in ("bh00", "albpow", "colh0", "temph0", "abu_c")], history=hist)
    finally:
        asora_core.device_close()
    np.savez(os.path.join(out_dir, f"evolve_{rank}.npz"), xh_new=xh_new, phi_ion=phi, niter=len(hist),
             conv_flags=np.array([h["conv_flag"] for h in hist]), conv_criterion=hist[0]["conv_criterion"])
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
