#!/usr/bin/env python3
"""Turn the rocprofv3 outputs of tools/gpu_bench_prof.sh (under gpurun_out/prof) into the two
summaries kept under profiles/:
  <tag>_bench_kernel_stats.csv  the --kernel-trace --stats table (per-kernel calls / avg / min / max)
  <tag>_bench_hbm_pmc.csv       per-kernel average FETCH_SIZE / WRITE_SIZE per dispatch, in bytes
                                (both counters tick in KB units; FETCH_SIZE x2 on gfx950 --
                                /opt/skills/guides/MI355X_MICROARCH.md, HBM section)
usage: summarize_prof.py <tag> [<prof dir under gpurun_out>]       e.g. r02_bench_scene_a r2k/prof"""
import csv, glob, os, sys, collections

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
prof = os.path.join(ROOT, "gpurun_out", sys.argv[2] if len(sys.argv) > 2 else "prof")


def one(pattern):
    f = sorted(glob.glob(os.path.join(prof, pattern), recursive=True))
    if not f:
        sys.exit(f"missing {pattern}")
    return f[-1]


# ---- kernel stats ---------------------------------------------------------------------------------
src = one("bench_kt/**/*kernel_stats.csv")
rows = list(csv.DictReader(open(src)))
keep = [r for r in rows if "rocprim" not in r["Name"] or True]
with open(os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.csv"), "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
    w.writeheader()
    for r in keep:
        r["Name"] = r["Name"][:100]
        w.writerow(r)


# ---- PMC ---------------------------------------------------------------------------------------------
def counter(dirname, name):
    src = one(f"{dirname}/**/*counter_collection.csv")
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(src)):
        if r["Counter_Name"] != name:
            continue
        a = acc[r["Kernel_Name"][:100]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return acc


fetch, write = counter("bench_fetch", "FETCH_SIZE"), counter("bench_write", "WRITE_SIZE")
with open(os.path.join(ROOT, "profiles", f"{tag}_hbm_pmc.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "dispatches", "FETCH_SIZE_avg_KB_raw", "fetch_bytes_avg_corrected_x2",
                "WRITE_SIZE_avg_KB_raw", "write_bytes_avg"])
    for k, (n, tot) in sorted(fetch.items(), key=lambda kv: -kv[1][1]):
        fk = tot / n
        wn, wt = write.get(k, [1, 0.0])
        wk = wt / max(1, wn)
        w.writerow([k, n, f"{fk:.1f}", int(fk * 1024 * 2), f"{wk:.1f}", int(wk * 1024)])
sha = os.path.join(prof, "csrc_sha16.txt")
import json
json.dump({"csrc_sha16": open(sha).read().strip() if os.path.exists(sha) else None,
           "note": "fingerprint (bench.csrc_sha16) of mp2p_icp_amd/csrc + include/mp2p_hip.h the counters were captured with"},
          open(os.path.join(ROOT, "profiles", f"{tag}_hbm_pmc.meta.json"), "w"))
print("written profiles/%s_kernel_stats.csv and profiles/%s_hbm_pmc.csv" % (tag, tag))
