"""Copies the data files the reference's own tests hold for this path into tests/golden/ (container only):

    tests/data/cinterp_output_{050505,010203,100515}.npy   (tests/test_asora_internal.py:10-20)
    tests/data/photo_ionization_rate.npy                   (tests/test_asora.py:100-109)

and writes the optical-depth grid of the reference's own ``make_tau_table`` (pyc2ray/radiation/common.py:11-35, a
numpy-only file that loads stand-alone) for the three table shapes the tests and the bench use.  Data only: inputs
and expected outputs.  Loaded with numpy.load(allow_pickle=False).

Note on the import below: it EXECUTES pyc2ray/radiation/common.py from the (untrusted) reference tree in this
interpreter.  That file was read in full before this script was first run: it imports numpy only, sets one constant and
defines ``make_tau_table`` (35 lines: ``10 ** (minlogtau + arange * dlogtau)`` behind a leading 0); no I/O, no
subprocess, no network.  The point of the .npz is to pin pyc2ray_amd.radiation.make_tau_table against the OUTPUT of the reference's
generator, which a restatement of those lines could not do.  Build container only; nothing on the GPU box runs this.
"""
import importlib.util
import os
import sys

import numpy as np

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))

for name in ("cinterp_output_050505", "cinterp_output_010203", "cinterp_output_100515", "photo_ionization_rate"):
    a = np.load(os.path.join(REF, "tests", "data", name + ".npy"), allow_pickle=False)
    np.save(os.path.join(HERE, "ref_" + name + ".npy"), a)
    print(name, a.shape, a.dtype)

spec = importlib.util.spec_from_file_location("ref_radiation_common", os.path.join(REF, "pyc2ray", "radiation", "common.py"))
mod = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mod)
out = {}
for key, (lo, hi, n) in {"m20_4_2000": (-20.0, 4.0, 2000), "m20_4_20000": (-20.0, 4.0, 20000), "m20_4_10000": (-20.0, 4.0, 10000)}.items():
    tau, dlogtau = mod.make_tau_table(lo, hi, n)
    out["tau_" + key] = np.asarray(tau, dtype=np.float64)
    out["dlogtau_" + key] = np.float64(dlogtau)
np.savez(os.path.join(HERE, "ref_tau_tables.npz"), **out)
print("tau tables", {k: v.shape for k, v in out.items()})
