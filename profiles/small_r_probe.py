"""Sweep-kernel time against the length of the source list at small radii (what a short launch costs).
usage: python profiles/small_r_probe.py [N] [R,R,...] [ns,ns,...]   (ASORA_OPTIONS=name=value,... applies)"""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pyc2ray_amd import libasora as asora
import bench

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
RS = [float(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "3,5,10").split(",")]
NS = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "256,1000,10000,100000").split(",")]
class A: pass
a = A(); a.N = N; a.numtau = 20000; a.heat = False; a.density = "uniform"; a.sources = max(NS); a.xh = 2e-4
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
for R in RS:
    visits, _ = asora.visit_count(R, N); ins = asora.geometry_census(R, dr, N)["owned_rate"]
    for ns in NS:
        asora.source_data_to_device(np.ascontiguousarray(inp["pos"][:ns].ravel()), np.ascontiguousarray(inp["flux"][:ns]), ns)
        ms = []
        for i in range(6):
            asora.do_all_sources_dev(R, 0, 6.30e-18, dr, xh.data_ptr(), phi.data_ptr(), 0, ns, N, -20.0, inp["dlogtau"], inp["thin"].shape[0], st)
            ms.append(asora.last_sweep_ms())
        k = float(np.median(ms[1:])); info = asora.last_launch_info()
        frac = (32.0 * visits + 16.0 * ins) * ns / (k * 1e-3) / 8e12
        print(json.dumps(dict(N=N, R=R, ns=ns, kernel_ms=round(k, 4), ns_per_source=round(k * 1e6 / ns, 2), frac=round(frac, 3), grid=info["grid"], block=info["block"],
                              lds=info["lds_bytes"], sect=info["sectors_per_workgroup"], pack=info["sources_per_workgroup"])), flush=True)
asora.device_close()
