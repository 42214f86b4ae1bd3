"""Per-call fixed cost (everything but the sweep kernel) of asora_do_all_sources_dev at 256^3: wall time per call minus
the sweep's HIP-event time, for a short list at a small radius.  usage: python profiles/fixed_cost_probe.py [N]"""
import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pyc2ray_amd import libasora as asora
import bench
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
class A: pass
a = A(); a.N = N; a.numtau = 20000; a.heat = False; a.density = "uniform"; a.sources = 10000; a.xh = 2e-4
inp = bench.make_inputs(a)
dev = torch.device("cuda", 0)
for kv in filter(None, os.environ.get("ASORA_OPTIONS", "").split(",")):
    k, v = kv.split("="); asora.set_option(k, int(v))
asora.device_init(N, 64, 0, 1)
asora.photo_table_to_device(np.ascontiguousarray(inp["thin"]), np.ascontiguousarray(inp["thick"]), inp["thin"].shape[0])
nd = torch.from_numpy(inp["ndens"].ravel()).to(dev); xh = torch.from_numpy(inp["xh"].ravel()).to(dev)
phi = torch.zeros(N ** 3, dtype=torch.float64, device=dev)
asora.density_from_device(nd.data_ptr(), N)
st = torch.cuda.current_stream().cuda_stream
dr = 3.0 * 3.0856775814913673e24 / 200
for R, ns in ((5.0, 1000), (10.0, 1000), (10.0, 10000)):
    asora.source_data_to_device(np.ascontiguousarray(inp["pos"][:ns].ravel()), np.ascontiguousarray(inp["flux"][:ns]), ns)
    call = lambda: asora.do_all_sources_dev(R, 0, 6.30e-18, dr, xh.data_ptr(), phi.data_ptr(), 0, ns, N, -20.0, inp["dlogtau"], inp["thin"].shape[0], st)
    for _ in range(3): call()
    torch.cuda.synchronize()
    K = 50
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(K): call()
    e1.record(); torch.cuda.synchronize()
    per_call = e0.elapsed_time(e1) / K
    k = asora.last_sweep_ms(); info = asora.last_launch_info()
    print(json.dumps(dict(N=N, R=R, sources=ns, stream_ms_per_call=round(per_call, 4), kernel_ms=round(k, 4), fixed_ms=round(per_call - k, 4),
                          one_layout=info["single_layout"])), flush=True)
asora.device_close()
