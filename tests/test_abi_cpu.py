"""CPU-side checks of the drop-in boundary: libnerfhip.so loads and exports every symbol that
include/nerfhip.h declares; the config struct mirrors the header; creating a handle without a GPU fails
loudly (no CPU fallback).  No compute call is made here."""
import ctypes as C
import os
import re

import pytest

from nerf_amd import NerfConfig, _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "nerfhip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(nerfhip_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = header_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"libnerfhip.so does not export {n}"
    assert set(names) == set(_lib.SYMBOLS), "ctypes table and header disagree"


def test_config_struct_matches_header():
    c = NerfConfig(skips=[4], NSamples=64, NImportance=128, maxRays=77, seed=5).to_c()
    assert (c.pos_l, c.dir_l, c.D, c.W, c.n_skips, c.skips[0]) == (10, 4, 8, 256, 1, 4)
    assert (c.n_samples, c.n_importance, c.max_rays, c.seed) == (64, 128, 77, 5)
    # sizeof(struct nerfhip_config): 26 4-byte fields (104 B, 8-aligned), uint64 seed, 2 ints
    assert C.sizeof(_lib.Config) == 120
    assert _lib.Config.seed.offset == 104


def _driver():
    path = os.path.join(ROOT, "tests", "bin", "c_abi_driver")
    assert os.path.exists(path), "tests/bin/c_abi_driver is missing: run __graft_entry__.build()"
    return path


def test_c_driver_layout_matches_ctypes():
    """The plain-C client (tests/c_abi_driver.c, gcc -std=c11 -pedantic against include/nerfhip.h) and the ctypes
    mirror agree on sizeof / offsetof of every struct of the ABI: what a JNI stub compiled from the header would see."""
    import subprocess
    out = subprocess.run([_driver(), "layout"], check=True, capture_output=True, text=True).stdout
    got = dict(l.rsplit(" ", 1) for l in out.strip().splitlines() if not l.startswith("version"))
    structs = {"nerfhip_config": _lib.Config, "nerfhip_aux": _lib.Aux, "nerfhip_prof": _lib.Prof}
    seen = 0
    for key, val in got.items():
        kind, name = key.split(".", 1)
        if kind == "sizeof":
            assert C.sizeof(structs[name]) == int(val), key
        elif kind in structs:
            assert getattr(structs[kind], name).offset == int(val), key
            seen += 1
        elif kind == "const":
            assert {"NERFHIP_MAX_SKIPS": _lib.MAX_SKIPS, "NERFHIP_DP_ID_BYTES": 128, "NERFHIP_E_COMM": -5}[name] == int(val)
    assert seen == sum(len(s._fields_) for s in structs.values())      # every field of every struct was compared
    assert "nerfhip" in out.splitlines()[-1]


def test_no_gpu_is_a_loud_error():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from nerf_amd import NerfBlock, NerfHipError
    with pytest.raises(NerfHipError):
        NerfBlock(NerfConfig())


def test_product_does_not_touch_the_oracle():
    """The product package must not import/link/call anything under oracle/ (task rule 3)."""
    pkg = os.path.join(ROOT, "nerf_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".inc", ".cpp", ".hpp")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in txt.lower(), f"{f} mentions the oracle"
