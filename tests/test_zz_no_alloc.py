"""asora_global_pass_dev_noalloc -- the one entry point that allocates nothing (include/asora_hip.h) -- through the
C-ABI against asora_global_pass_dev, the entry point it is a variant of: same kernel, reduction record in a
file-scope device array instead of a hipMalloc'ed one.  Two calls in a row, the second on a SMALLER mesh (the
scratch keeps the last call's contents: what must start at zero is cleared by the call).  Bar: the two grids the
pass updates bit-identical, the count of non-converged cells equal, the two sums (f64 atomics, order-free) to 1e-13.
A second test holds the entry point to what its header promises: the device's free memory (hipMemGetInfo) does not
move across repeated calls, and the form usable under graph capture (no results wanted: no copy, no wait) leaves
the same grids."""
import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu


def run_pass(asora, c, noalloc):
    import torch
    dev = torch.device("cuda", asora.device_index())
    t = {k: torch.from_numpy(np.ascontiguousarray(c[k]).ravel().copy()).to(dev)
         for k in ("ndens", "temp", "xh", "xh_av", "xh_intermed", "phi_ion", "clump")}
    stats = asora.global_pass_dev(c["dt"], t["ndens"].data_ptr(), t["temp"].data_ptr(), t["xh"].data_ptr(),
                                  t["xh_av"].data_ptr(), t["xh_intermed"].data_ptr(), t["phi_ion"].data_ptr(),
                                  t["clump"].data_ptr(), c["bh00"], c["albpow"], c["colh0"], c["temph0"], c["abu_c"],
                                  c["N"], torch.cuda.current_stream(dev).cuda_stream, noalloc=noalloc)
    return stats, t["xh_av"].cpu().numpy(), t["xh_intermed"].cpu().numpy()


def test_noalloc_chemistry_pass_equals_the_allocating_entry_point():
    from pyc2ray_amd import libasora as asora
    big, small = cases.chemistry_case(N=40, seed=3), cases.chemistry_case(N=17, seed=8)
    asora.device_init(40, 8, 0, 1)
    try:
        for c in (big, small):  # second call: smaller input, same scratch
            ref_stats, ref_av, ref_int = run_pass(asora, c, noalloc=False)
            stats, av, inter = run_pass(asora, c, noalloc=True)
            assert np.array_equal(av, ref_av) and np.array_equal(inter, ref_int)
            assert stats[0] == ref_stats[0] and 0 < stats[0] < c["N"] ** 3
            assert stats[1] == pytest.approx(ref_stats[1], rel=1e-13) and stats[2] == pytest.approx(ref_stats[2], rel=1e-13)
            assert stats[1] == pytest.approx(float(inter.sum()), rel=1e-12)
        # sizes beyond the stated maximum are refused with a clear error, before anything is launched
        with pytest.raises(RuntimeError, match="exceeds the maximum of 1024"):
            asora.global_pass_dev(1.0, 8, 8, 8, 8, 8, 8, 8, 1.0, 1.0, 1.0, 1.0, 1.0, 1025, noalloc=True)
    finally:
        asora.device_close()


def test_noalloc_entry_point_allocates_nothing_and_runs_without_results():
    import torch
    from pyc2ray_amd import libasora as asora
    c = cases.chemistry_case(N=40, seed=3)
    keys = ("ndens", "temp", "xh", "xh_av", "xh_intermed", "phi_ion", "clump")
    asora.device_init(40, 8, 0, 1)
    try:
        dev = torch.device("cuda", asora.device_index())
        stream = torch.cuda.current_stream(dev).cuda_stream
        host = {k: torch.from_numpy(np.ascontiguousarray(c[k]).ravel().copy()) for k in keys}
        t = {k: v.to(dev) for k, v in host.items()}
        u = {k: v.to(dev) for k, v in host.items()}
        chem = [c[k] for k in ("bh00", "albpow", "colh0", "temph0", "abu_c")]
        ptrs = lambda d: [d[k].data_ptr() for k in keys]
        # warm every lazy path of the runtime (kernel load, stream, the page-locked landing buffer made at init)
        asora.global_pass_dev(c["dt"], *ptrs(u), *chem, c["N"], stream, noalloc=True)
        for k in ("xh_av", "xh_intermed"):
            u[k].copy_(host[k])
        torch.cuda.synchronize(dev)
        free0, _ = torch.cuda.mem_get_info(dev)   # hipMemGetInfo: bytes free on the device, all processes' allocations
        for _ in range(20):
            stats = asora.global_pass_dev(c["dt"], *ptrs(t), *chem, c["N"], stream, noalloc=True)
        # the form for graph capture / serving loops: no results wanted -> memset + kernel on the stream, no copy, no wait
        assert asora.global_pass_dev(c["dt"], *ptrs(u), *chem, c["N"], stream, want_stats=False, noalloc=True) is None
        torch.cuda.synchronize(dev)
        free1, _ = torch.cuda.mem_get_info(dev)
        assert free1 == free0, f"device memory moved by {free0 - free1} bytes across 21 no-alloc calls"
        assert 0 < stats[0] < c["N"] ** 3
        # one pass without results == one pass with results (t has seen 20 passes: compare u with a fresh single pass)
        w = {k: v.to(dev) for k, v in host.items()}
        asora.global_pass_dev(c["dt"], *ptrs(w), *chem, c["N"], stream, noalloc=False)
        for k in ("xh_av", "xh_intermed"):
            assert torch.equal(u[k], w[k]), k
    finally:
        asora.device_close()


def test_overlapping_calls_take_their_own_reduction_record():
    """The chemistry kernel's reduction record comes from a ring of 8 (include/asora_hip.h): a call that asks for its
    results while another call -- issued without results on another stream, here on a 64x larger mesh, so it is
    still running -- reduces into the scratch must still get its own counts and sums."""
    import torch
    from pyc2ray_amd import libasora as asora
    big, small = cases.chemistry_case(N=96, seed=5), cases.chemistry_case(N=24, seed=6)
    keys = ("ndens", "temp", "xh", "xh_av", "xh_intermed", "phi_ion", "clump")
    chem = lambda c: [c[k] for k in ("bh00", "albpow", "colh0", "temph0", "abu_c")]
    asora.device_init(96, 8, 0, 1)
    try:
        dev = torch.device("cuda", asora.device_index())
        up = lambda c: {k: torch.from_numpy(np.ascontiguousarray(c[k]).ravel().copy()).to(dev) for k in keys}
        ptrs = lambda d: [d[k].data_ptr() for k in keys]
        ref = asora.global_pass_dev(small["dt"], *ptrs(up(small)), *chem(small), small["N"],
                                    torch.cuda.current_stream(dev).cuda_stream, noalloc=True)
        side = torch.cuda.Stream(dev)
        tb, ts = up(big), up(small)
        torch.cuda.synchronize(dev)
        for _ in range(3):  # three calls in flight on the side stream, none asks for results
            assert asora.global_pass_dev(big["dt"], *ptrs(tb), *chem(big), big["N"], side.cuda_stream, want_stats=False,
                                         noalloc=True) is None
        got = asora.global_pass_dev(small["dt"], *ptrs(ts), *chem(small), small["N"],
                                    torch.cuda.current_stream(dev).cuda_stream, noalloc=True)
        torch.cuda.synchronize(dev)
        assert got[0] == ref[0] and got[1] == pytest.approx(ref[1], rel=1e-13) and got[2] == pytest.approx(ref[2], rel=1e-13)
    finally:
        asora.device_close()
