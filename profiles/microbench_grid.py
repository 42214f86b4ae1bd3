"""The reference's micro-benchmark grid (tests/test_asora.py:112-121: mesh 32 / 64 / 128 x 100 / 1000 / 10000 sources, R = 15,
cell 50 pc / mesh, the host-pointer call asora.do_all_sources timed after one warm-up call), run on this library.
The reference's third axis, batch_size 8 / 16 / 32, does not exist here (num_src_par is accepted and ignored).
One JSON line per point: wall time of the call (what pytest-benchmark times: H2D xh_av + sweep + D2H phi_ion and coldensh_out),
the sweep kernel's HIP-event time inside it, and the launch the library picked.
usage: python profiles/microbench_grid.py > profiles/rN_microbench_grid.jsonl"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases
from pyc2ray_amd import libasora as asora

for mesh in (32, 64, 128):
    for ns in (100, 1000, 10000):
        c = cases.reference_test_case(N=mesh, ns=ns, R=15.0)
        c["dr"] = 50 * cases.PC / mesh
        asora.device_init(mesh, 8, 0, 1)
        try:
            asora.photo_table_to_device(c["thin"], c["thick"], c["num_tau"])
            asora.density_to_device(np.ravel(c["ndens"]).copy(), mesh)
            asora.source_data_to_device(np.ascontiguousarray(c["pos_flat"]), c["flux"], ns)
            cdh, phi, xh = np.zeros(mesh ** 3), np.zeros(mesh ** 3), np.ravel(c["xh"]).copy()
            call = lambda: asora.do_all_sources(c["R"], cdh, c["sig"], c["dr"], None, xh, phi, ns, mesh, c["minlogtau"],
                                                c["dlogtau"], c["num_tau"])
            call()  # the reference's manual warm-up
            wall, kern = [], []
            for _ in range(7):
                t0 = time.perf_counter(); call(); wall.append(time.perf_counter() - t0); kern.append(asora.last_sweep_ms())
            info = asora.last_launch_info()
        finally:
            asora.device_close()
        visits, _ = asora.visit_count(c["R"], mesh)
        print(json.dumps(dict(mesh=mesh, sources=ns, R=15.0, call_ms=round(float(np.median(wall)) * 1e3, 4),
                              kernel_ms=round(float(np.median(kern)), 4), visits_per_source=visits,
                              updates_per_s_call=visits * ns / float(np.median(wall)),
                              launch=f"{info['block']} thr, {info['staging']}, {info['sectors_per_workgroup']} sector(s) x "
                                     f"{info['sources_per_workgroup']} source(s)" + (", one layout" if info["single_layout"] else ""),
                              phi_max=float(phi.max()))), flush=True)
