"""Builds libmp2p_hip.so (the C-ABI drop-in, include/mp2p_hip.h) for gfx950 with hipcc.

hipcc cross-compiles without a GPU; the .so is written in-tree (mp2p_icp_amd/libmp2p_hip.so)
so that it travels to the GPU box with the repository snapshot.
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC_DIR = os.path.join(HERE, "csrc")
UNITY = os.path.join(SRC_DIR, "mp2p_hip_all.hip")
LIB = os.path.join(HERE, "libmp2p_hip.so")

# -ffp-contract=off: the fp32 distance / threshold / transform expressions must round exactly
# like the reference's x86-64 build (no FMA); see csrc/device_utils.hpp.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
         "-Wno-unused-result"]


def sources():
    out = [os.path.join(HERE, "..", "include", "mp2p_hip.h")]
    for f in sorted(os.listdir(SRC_DIR)):
        if f.endswith((".hip", ".hpp")):
            out.append(os.path.join(SRC_DIR, f))
    return out


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(s) > t for s in sources())


def hipcc_path():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def build(force=False, verbose=False):
    """Compile the HIP library if missing or stale.  Returns the path of the .so."""
    if not force and not needs_build():
        return LIB
    cc = hipcc_path()
    if not os.path.exists(cc):
        raise RuntimeError("hipcc not found: cannot build libmp2p_hip.so")
    cmd = [cc] + FLAGS + [UNITY, "-o", LIB + ".tmp"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + r.stderr[-8000:])
    os.replace(LIB + ".tmp", LIB)
    if verbose:
        print("built", LIB)
    return LIB


# ---- the host-path library: adapter/mp2p_hip_host.hpp (the MRPT-free part of the reference-side
#      plugin) behind a C ABI, for tests/ and bench.py.  Plain g++, links libmp2p_hip.so. ------------
ADAPTER_DIR = os.path.join(HERE, "..", "adapter")
HOSTPATH_SRC = os.path.join(ADAPTER_DIR, "hostpath_capi.cpp")
HOSTPATH_LIB = os.path.join(HERE, "libmp2p_hip_hostpath.so")


def hostpath_needs_build():
    if not os.path.exists(HOSTPATH_LIB):
        return True
    t = os.path.getmtime(HOSTPATH_LIB)
    deps = [HOSTPATH_SRC, os.path.join(ADAPTER_DIR, "mp2p_hip_host.hpp"),
            os.path.join(HERE, "..", "include", "mp2p_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build_hostpath(force=False, verbose=False):
    if not force and not hostpath_needs_build():
        return HOSTPATH_LIB
    build()  # links against the HIP library
    cxx = shutil.which("g++") or "g++"
    cmd = [cxx, "-O2", "-std=c++17", "-shared", "-fPIC", "-Wall", HOSTPATH_SRC,
           "-I" + os.path.join(HERE, "..", "include"), "-I" + ADAPTER_DIR, "-L" + HERE, "-lmp2p_hip",
           "-Wl,-rpath,$ORIGIN", "-lpthread", "-ldl", "-o", HOSTPATH_LIB + ".tmp"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("g++ failed on the host-path library:\n" + r.stderr[-8000:])
    os.replace(HOSTPATH_LIB + ".tmp", HOSTPATH_LIB)
    if verbose:
        print("built", HOSTPATH_LIB)
    return HOSTPATH_LIB


if __name__ == "__main__":
    build(force=True, verbose=True)
    build_hostpath(force=True, verbose=True)
