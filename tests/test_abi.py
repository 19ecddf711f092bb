"""CPU tests of the drop-in boundary: the C-ABI library loads, exports exactly what
include/ntcard_hip.h declares, and refuses to run without a device (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import orc
from ntcard_amd import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "ntcard_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ntc_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    L = _abi.lib()
    declared = header_symbols()
    assert declared, "no declarations parsed from the header"
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/ntcard_hip.h but not exported"
    assert sorted(_abi.ABI_SYMBOLS) == declared
    assert L.ntc_abi_version() == 6
    assert L.ntc_max_k() >= 128


def test_header_cites_reference_seam():
    text = open(os.path.join(ROOT, "include", "ntcard_hip.h")).read()
    for cite in ("ntcard.cpp:147-171", "ntcard.cpp:437-439", "ntcard.cpp:237-275"):
        assert cite in text


def test_estimator_matches_oracle_and_golden(golden_dir):
    """ntc_estimate / ntc_write_hist are host code: run here, check against the oracle + reference"""
    import json
    import ntcard_amd as nt
    with open(os.path.join(golden_dir, "sketch_goldens.json")) as f:
        gold = json.load(f)
    for ent in gold:
        for pl in ent["planes"]:
            p = np.zeros((2, 65536), dtype=np.uint32)
            for s, v, c in pl["p_nonzero"]:
                p[s, v] = c
            F0, f = nt.estimate(p, ent["r_bits"], ent["s_bits"], 1000)
            assert F0 == pl["F0"]
            assert f[1:].tolist() == pl["f_1_1000"]
            oF0, of = orc.comp_est_p(p, ent["r_bits"], ent["s_bits"], 1000)
            assert F0 == oF0 and np.array_equal(f, of[:1001])
            F0c, fc = nt.estimate(p, ent["r_bits"], ent["s_bits"], 37)
            assert F0c == F0 and np.array_equal(fc, f[:38])


def test_write_hist_format(tmp_path, golden_dir):
    import json
    import ntcard_amd as nt
    with open(os.path.join(golden_dir, "sketch_goldens.json")) as f:
        ent = [e for e in json.load(f) if e["r_bits"] == 27][0]
    pl = ent["planes"][0]
    p = np.zeros((2, 65536), dtype=np.uint32)
    for s, v, c in pl["p_nonzero"]:
        p[s, v] = c
    F0, f = nt.estimate(p, 27, 7, 1000)
    out = tmp_path / "x_k12.hist"
    nt.write_hist(out, ent["f1"][0], F0, f, 1000)
    assert out.read_bytes() == open(os.path.join(golden_dir, "ref_k12__out_k12.hist"), "rb").read()


def test_no_cpu_fallback():
    """Without a HIP device the engine must fail loudly, never compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    L = _abi.lib()
    k = (C.c_uint32 * 1)(32)
    cfg = _abi.NtcConfig(n_k=1, k=C.cast(k, C.POINTER(C.c_uint32)), gap=0, r_bits=20, s_bits=7, device=0)
    h = C.c_void_p()
    rc = L.ntc_create(C.byref(cfg), C.byref(h))
    assert rc == -2 and not h.value
    assert b"no HIP device" in L.ntc_last_error() or b"failed" in L.ntc_last_error()


def test_bad_arguments_are_rejected_before_touching_the_device():
    L = _abi.lib()
    h = C.c_void_p()
    k = (C.c_uint32 * 2)(32, 0)
    cfg = _abi.NtcConfig(n_k=2, k=C.cast(k, C.POINTER(C.c_uint32)), gap=0, r_bits=27, s_bits=7, device=0)
    assert L.ntc_create(C.byref(cfg), C.byref(h)) == -1  # k = 0
    cfg = _abi.NtcConfig(n_k=0, k=C.cast(k, C.POINTER(C.c_uint32)), gap=0, r_bits=27, s_bits=7, device=0)
    assert L.ntc_create(C.byref(cfg), C.byref(h)) == -1
    k1 = (C.c_uint32 * 1)(12)
    cfg = _abi.NtcConfig(n_k=1, k=C.cast(k1, C.POINTER(C.c_uint32)), gap=3, r_bits=27, s_bits=7, device=0)
    assert L.ntc_create(C.byref(cfg), C.byref(h)) == -1  # g%2 != k%2 (ntcard.cpp:382-385)
    assert L.ntc_create(None, C.byref(h)) == -1
    for gone in (4, 256, 1 << 20):  # ABI 4's NTC_FLAG_BITSLICE_KERNEL / NTC_FLAG_TILED_TEAMS selected kernels that were retired; unknown bits are an error, not ignored
        cfg = _abi.NtcConfig(n_k=1, k=C.cast(k1, C.POINTER(C.c_uint32)), gap=0, r_bits=20, s_bits=7, device=0, flags=gone)
        assert L.ntc_create(C.byref(cfg), C.byref(h)) == -1 and b"flag" in L.ntc_last_error()
