"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads and exports every
symbol include/mp2p_hip.h declares; the product never touches the oracle; and without a GPU
every compute entry fails loudly (no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "mp2p_hip.h")


def _declared_symbols():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mp2p_hip_[a-z0-9_]+)\s*\(", txt)))


def test_library_builds_and_exports_every_declared_symbol():
    from mp2p_icp_amd import _build, _lib
    so = _build.build()
    assert os.path.exists(so)
    L = C.CDLL(so)
    declared = _declared_symbols()
    assert len(declared) >= 40
    for name in declared:
        assert hasattr(L, name), f"{name} declared in mp2p_hip.h but not exported"
    # the ctypes table covers exactly the header
    assert sorted(_lib.SIGNATURES) == declared
    assert L.mp2p_hip_abi_version() == 1


def test_struct_layouts_match_header():
    from mp2p_icp_amd import _lib
    assert C.sizeof(_lib.GNResult) == 12 * 8 + 36 * 8 + 6 * 8 + 8 + 8
    assert _lib.PAIR_PT2PT.itemsize == 36 and _lib.PAIR_PT2PL.itemsize == 72
    # cross-check against the C compiler's view of the header
    src = r'''
#include <stdio.h>
#include "mp2p_hip.h"
int main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(mp2p_hip_map_params),
 sizeof(mp2p_hip_map_info), sizeof(mp2p_hip_pt2pt_params), sizeof(mp2p_hip_pt2pl_params),
 sizeof(mp2p_hip_gn_params), sizeof(mp2p_hip_gn_result), sizeof(mp2p_hip_stats),
 sizeof(mp2p_hip_pair_pt2pl)); return 0;}'''
    d = os.path.join(ROOT, "tests", "_tmp_abi")
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "s.c"), "w") as f:
        f.write(src)
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "s.c"), "-o",
                           os.path.join(d, "s")])
    got = [int(x) for x in subprocess.check_output([os.path.join(d, "s")]).split()]
    want = [C.sizeof(_lib.MapParams), C.sizeof(_lib.MapInfo), C.sizeof(_lib.Pt2PtParams),
            C.sizeof(_lib.Pt2PlParams), C.sizeof(_lib.GNParams), C.sizeof(_lib.GNResult),
            C.sizeof(_lib.Stats), 72]
    assert got == want


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "mp2p_icp_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f
                assert "mp2p_oracle" not in txt or f.endswith((".hip", ".hpp")) and "oracle/mp2p_oracle.c" in txt, f
                assert "libmp2p_oracle" not in txt, f


def test_nn_kernels_have_no_fma_in_distance_math():
    """bit-exact indices need separately rounded mul/add (device_utils.hpp): the ISA of the
    search kernels must not contain fp32/fp64 fused multiply-adds."""
    from mp2p_icp_amd import _build
    so = _build.build()
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump not available")
    d = os.path.join(ROOT, "tests", "_tmp_abi")
    os.makedirs(d, exist_ok=True)
    # extract the gfx950 code object bundled in the .so (llvm-objdump writes next to its input)
    import glob
    import shutil
    cp = os.path.join(d, "lib.so")
    shutil.copy(so, cp)
    for f in glob.glob(cp + ".*"):
        os.remove(f)
    subprocess.run([objdump, "--offloading", cp], capture_output=True)
    cos = glob.glob(cp + ".*gfx950*")
    if not cos:
        pytest.skip("cannot extract the device code object")
    co = cos[0]
    asm = subprocess.check_output([objdump, "-d", co], text=True)
    cur, bad = None, []
    for line in asm.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
        if m:
            cur = m.group(1)
            continue
        if not (cur and "nn_tile_kernel" in cur):
            continue
        m = re.search(r"\b(v_(?:fma|fmac|mad|mac|pk_fma)_(?:f32|f64|legacy_f32)\S*)\s+(.*?)\s*//", line)
        if not m:
            continue
        op, args = m.group(1), m.group(2)
        # the only legitimate fused ops are inside the correctly-rounded sqrtf expansion
        # (v_fma_f32 d, -a, b, c) and the u64-division expansion (v_fmac_f32 with the
        # literals 2^32 / -2^32 / 0); anything else would be a contracted distance term
        if op.startswith("v_fma_f32") and re.match(r"^v\d+, -v\d+, v\d+, v\d+$", args):
            continue
        if op.startswith("v_fmac_f32") and re.search(r", (0x4f800000|0xcf800000|0), v\d+$", args):
            continue
        bad.append((cur, line.strip()))
    assert not bad, bad[:5]


def test_no_device_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import mp2p_icp_amd
    with pytest.raises(mp2p_icp_amd.Mp2pHipError) as e:
        mp2p_icp_amd.Context(0)
    assert e.value.code == -5 and "no CPU fallback" in str(e.value)
