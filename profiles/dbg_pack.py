import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from pyc2ray_amd import libasora as asora
from test_gpu_parity import run_hip, load_golden
c = load_golden("tests/golden", "rcut7p3")
asora.set_option("quad_kernel", -1)
ref = {n: run_hip(asora, c, num_src=n) for n in (1, 2)}
for qk, sp in ((2, 3),):
    asora.set_option("quad_kernel", qk); asora.set_option("src_pack", sp)
    for n in (1,):
        out = run_hip(asora, c, num_src=n)
        info = asora.last_launch_info()
        d = out[1] != ref[n][1]
        print(qk, sp, n, info["sectors_per_workgroup"], info["sources_per_workgroup"], info["grid"], info["lds_bytes"], "cdh diff cells:", int(d.sum()),
              "phi maxrel:", float(np.max(np.abs(out[0] - ref[n][0]) / np.maximum(ref[n][0], 1e-300))))
        if d.any():
            idx = np.argwhere(d)
            src = c["pos"][n - 1]
            off = (idx - src + 16) % 32 - 16
            print("   offsets of differing cells (first 12):", off[:12].tolist())
            print("   values:", out[1][d][:6], ref[n][1][d][:6])
