"""Instruction mix of the largest loop of one kernel in a -save-temps .s file.
    python tools/isa_mix.py <file.s> <mangled-name-substring>"""
import collections
import re
import sys

src = open(sys.argv[1]).read().split("\n")
start = next(i for i, l in enumerate(src) if l.startswith("_Z") and sys.argv[2] in l and ":" in l.split(";")[0])
end = next(i for i in range(start, len(src)) if "s_endpgm" in src[i])
lines = [l.strip() for l in src[start:end + 1]]
labels = {}
for i, l in enumerate(lines):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        labels[m.group(1)] = i
best = None
for i, l in enumerate(lines):
    m = re.match(r"^s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
    if m and labels.get(m.group(1), 1 << 30) < i:
        span = i - labels[m.group(1)]
        if best is None or span > best[0]:
            best = (span, labels[m.group(1)], i)
cnt = collections.Counter()
for l in lines[best[1]:best[2]]:
    if not l or l[0] in ".;" or l.endswith(":"):
        continue
    cnt[l.split()[0]] += 1
groups = collections.Counter()
for op, n in cnt.items():
    key = ("mfma" if op.startswith("v_mfma") else "valu" if op.startswith("v_") else "lds" if op.startswith("ds_")
           else "vmem" if op.split("_")[0] in ("global", "buffer", "flat", "scratch") else "salu" if op.startswith("s_") else "other")
    groups[key] += n
print("loop of", best[0], "lines;", sum(cnt.values()), "instructions:", dict(groups))
for op, n in cnt.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 40):
    print(f"  {op:34s}{n}")
