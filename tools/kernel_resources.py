#!/usr/bin/env python3
"""Summarise hipcc -Rpass-analysis=kernel-resource-usage for our kernels (name, SGPR, VGPR,
scratch, occupancy, LDS)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "mp2p_icp_amd", "csrc", "mp2p_hip_all.hip")


def main():
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
           "-c", "-Wno-unused-result", SRC, "-o", "/tmp/_kr.o",
           "-Rpass-analysis=kernel-resource-usage"]
    out = subprocess.run(cmd, capture_output=True, text=True).stderr
    cur = None
    rows = {}
    for line in out.splitlines():
        m = re.search(r"remark: .*?Function Name: (\S+)", line) or re.search(r"remark:\s+Name: (\S+)", line)
        if m:
            cur = m.group(1)
            rows[cur] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+(?:\[[^\]]*\])?[A-Za-z /]*): (\d+)", line)
        if m and cur:
            rows[cur][m.group(1).strip()] = int(m.group(2))
    print(f"{'kernel':60s} {'SGPR':>5s} {'VGPR':>5s} {'AGPR':>5s} {'scr':>5s} {'occ':>4s} {'LDS':>6s}")
    for k, v in rows.items():
        if "rocprim" in k or "hipcub" in k:
            continue
        name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(.*", "", name)[:60]
        print(f"{name:60s} {v.get('TotalSGPRs', -1):5d} {v.get('VGPRs', -1):5d} {v.get('AGPRs', -1):5d} "
              f"{v.get('ScratchSize [bytes/lane]', -1):5d} {v.get('Occupancy [waves/SIMD]', -1):4d} "
              f"{v.get('LDS Size [bytes/block]', -1):6d}")


if __name__ == "__main__":
    main()
