"""Static instruction mix of one kernel in a device assembly file (hipcc --cuda-device-only -S):
python tools/isa_mix.py all.s _ZN4mp2p14nn_tile_kernelILi32ELb0EEEvNS_6NNArgsE"""
import collections
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
sym = sys.argv[2] + ":"
start = next(i for i, l in enumerate(lines) if l.startswith(sym))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
cnt = collections.Counter()
canon = 0
for l in lines[start:end + 1]:
    t = l.strip()
    if not t or t[0] in ";." or t.endswith(":"):
        continue
    cnt[t.split()[0]] += 1
    m = re.match(r"v_max_f32_e(32|64) (v\d+), (\S+), (\S+)", t)
    if m and m.group(3) == m.group(4):
        canon += 1
cls = collections.Counter()
for k, v in cnt.items():
    c = ("valu" if k.startswith("v_") else "salu" if k.startswith("s_") else "lds" if k.startswith("ds_")
         else "vmem" if k.startswith(("global_", "buffer_", "scratch_", "flat_")) else "other")
    cls[c] += v
print("static instructions:", sum(cnt.values()), dict(cls), "canonicalising v_max x,x:", canon)
for k, v in cnt.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 25):
    print(f"  {k:28s}{v}")
