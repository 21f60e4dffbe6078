#!/usr/bin/env python3
"""Static instruction counts of ONE kernel attributed to source lines (hipcc -S -gline-tables-only; CPU only).
usage: isa_by_line.py <mangled kernel name or a substring of it> [file substring = nn_seltile.hip] [bucket = 25 lines]
Prints, per bucket of source lines of the chosen file, the instructions (and the vector ones) the kernel's listing holds for it,
and the totals per source file (inlined helpers land in their own files).  Multiply by trip counts for a dynamic estimate:
DESIGN.md section 4, "Where a tile's instructions go"."""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "mp2p_icp_amd", "csrc", "mp2p_hip_all.hip")


def main():
    want = sys.argv[1] if len(sys.argv) > 1 else "nn_seltile_kernelILb0ELb0ELb0ELi4ELi0"
    fsub = sys.argv[2] if len(sys.argv) > 2 else "nn_seltile.hip"
    bucket = int(sys.argv[3]) if len(sys.argv) > 3 else 25
    out = "/tmp/_isa_by_line.s"
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-S", "--cuda-device-only",
                    "-gline-tables-only", "-Wno-unused-result", SRC, "-o", out], check=True, capture_output=True)
    text = open(out).read().split("\n")
    files = {}
    for l in text:
        m = re.match(r'\s*\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', l) or re.match(r'\s*\.file\s+(\d+)\s+"([^"]+)"', l)
        if m:
            files[int(m.group(1))] = os.path.basename(m.group(2))
    start = next((i for i, l in enumerate(text) if re.match(r"^_Z\w*:", l) and want in l), None)
    if start is None:
        sys.exit(f"no kernel whose mangled name holds {want!r}")
    cur, cnt, valu = None, collections.Counter(), collections.Counter()
    for l in text[start + 1:]:
        t = l.strip()
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", t)
        if m:
            cur = (int(m.group(1)), int(m.group(2)))
            continue
        if t.startswith("s_endpgm"):
            break
        if not t or t.startswith((".", ";")) or t.endswith(":"):
            continue
        cnt[cur] += 1
        if t.split()[0].startswith("v_"):
            valu[cur] += 1
    print(text[start].rstrip(":"))
    byfile = collections.Counter()
    for (f, _), v in cnt.items():
        byfile[files.get(f, str(f))] += v
    print("instructions per source file:", dict(byfile.most_common()))
    fid = [k for k, v in files.items() if fsub in v]
    rows = collections.defaultdict(lambda: [0, 0])
    for (f, ln), v in cnt.items():
        if f in fid:
            rows[ln // bucket][0] += v
            rows[ln // bucket][1] += valu[(f, ln)]
    print(f"{fsub}: lines -> instructions (vector)")
    for b in sorted(rows):
        print(f"  {b * bucket:5d}-{b * bucket + bucket - 1:5d}  {rows[b][0]:5d}  ({rows[b][1]})")


if __name__ == "__main__":
    main()
