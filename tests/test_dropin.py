"""The drop-in recipe of INTEGRATION.md section 1, executed: ``libasora_hip.so`` and ``libasora.py`` copied into a
``pyc2ray/lib/``-shaped package (where the reference's meson build installs its extension,
src/asora/meson.build:39-64), loaded the way the reference loads it (``from .lib import libasora``,
pyc2ray/load_extensions.py:35-54), called in the order of pyc2ray/asora_core.py:38,61 and
pyc2ray/evolve.py:192,201,254-267.  Runs in a child process so that the temporary package's module names
cannot meet this repository's."""
import os
import shutil
import subprocess
import sys
import textwrap

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

LOADER = '''
_asora = None


def load_asora():
    global _asora
    if _asora is None:
        from .lib import libasora
        _asora = libasora
    return _asora
'''


def make_package(tmp_path):
    from pyc2ray_amd import libasora
    lib = tmp_path / "site" / "pyc2ray" / "lib"
    lib.mkdir(parents=True)
    (tmp_path / "site" / "pyc2ray" / "__init__.py").write_text("")
    (tmp_path / "site" / "pyc2ray" / "load_extensions.py").write_text(LOADER)
    (lib / "__init__.py").write_text("")
    shutil.copy(os.path.join(ROOT, "pyc2ray_amd", "libasora.py"), lib / "libasora.py")
    shutil.copy(libasora.LIB_PATH, lib / "libasora_hip.so")
    return str(tmp_path / "site")


def run_child(site, code, **kw):
    env = dict(os.environ, PYTHONPATH=site)
    env.pop("ASORA_HIP_LIB", None)
    return subprocess.run([sys.executable, "-c", textwrap.dedent(code)], capture_output=True, text=True, env=env,
                          cwd=site, timeout=300, **kw)


def test_shim_loads_from_a_pyc2ray_lib_package(tmp_path):
    site = make_package(tmp_path)
    r = run_child(site, '''
        import os
        from pyc2ray.load_extensions import load_asora
        la = load_asora()
        assert la.__name__ == "pyc2ray.lib.libasora"
        here = os.path.dirname(os.path.abspath(la.__file__))
        assert la.LIB_PATH == os.path.join(here, "libasora_hip.so"), la.LIB_PATH   # next to the shim, no lib/lib
        for f in ("device_init", "device_close", "density_to_device", "photo_table_to_device",
                  "source_data_to_device", "do_all_sources"):
            assert callable(getattr(la, f))
        assert la.visit_count(60, 256)[0] == 1521729      # a host-only call through the copied library
        print("dropin-ok")
    ''')
    assert r.returncode == 0 and "dropin-ok" in r.stdout, r.stderr[-2000:]


@pytest.mark.gpu
def test_six_calls_through_the_dropped_in_module(tmp_path):
    import cases
    from test_gpu_parity import run_hip
    from pyc2ray_amd import libasora
    site = make_package(tmp_path)
    c = cases.case_multi16_unequal()
    np.savez(tmp_path / "case.npz", **{k: v for k, v in c.items() if v is not None})
    r = run_child(site, f'''
        import numpy as np
        from pyc2ray.load_extensions import load_asora
        libasora = load_asora()
        c = np.load({str(tmp_path / "case.npz")!r})
        N, NumSrc, NumTau = int(c["N"]), len(c["flux"]), int(c["num_tau"])
        libasora.device_init(N, 8, 0, 1)                                          # asora_core.py:38
        libasora.photo_table_to_device(c["thin"], c["thick"], NumTau)             # asora_core.py:61
        srcpos_flat = np.ravel(c["pos"].T.astype("int32", copy=True), order="F")  # sourceutils.py:30
        normflux_flat = c["flux"].astype("float64", copy=True)
        libasora.source_data_to_device(srcpos_flat, normflux_flat, NumSrc)        # evolve.py:192
        coldensh_out_flat = np.ravel(np.zeros((N, N, N), dtype="float64"))
        phi_ion_flat = np.ravel(np.zeros((N, N, N), dtype="float64"))
        ndens_flat = np.ravel(c["ndens"]).astype("float64", copy=True)
        xh_av_flat = np.ravel(c["xh"]).astype("float64", copy=True)
        libasora.density_to_device(ndens_flat, N)                                 # evolve.py:201
        ret = libasora.do_all_sources(float(c["R"]), coldensh_out_flat, float(c["sig"]), float(c["dr"]), ndens_flat,
                                      xh_av_flat, phi_ion_flat, NumSrc, N, float(c["minlogtau"]),
                                      float(c["dlogtau"]), NumTau)                # evolve.py:254-267
        assert ret is None
        libasora.device_close()
        np.save({str(tmp_path / "phi.npy")!r}, phi_ion_flat)
        np.save({str(tmp_path / "cdh.npy")!r}, coldensh_out_flat)
        print("dropin-ok")
    ''')
    assert r.returncode == 0 and "dropin-ok" in r.stdout, r.stderr[-2000:]
    phi, cdh, _ = run_hip(libasora, c)
    assert np.array_equal(np.load(tmp_path / "cdh.npy"), cdh.ravel())
    assert np.allclose(np.load(tmp_path / "phi.npy"), phi.ravel(), rtol=1e-13, atol=0)
