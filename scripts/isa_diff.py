"""Compare the kernels of two hipcc assembly files function by function (labels normalised, comments dropped): the check that
a source refactor or an added variant left the ISA of the existing kernels untouched.
usage: hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -I deepsvg_amd/csrc <file>.hip -o new.s   (twice)
       python scripts/isa_diff.py old.s new.s"""
import re
import sys


def funcs(path):
    out, cur = {}, None
    for line in open(path).read().split("\n"):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1)
            out[cur] = []
            continue
        if line.startswith(".Lfunc_end"):
            cur = None
        if cur:
            t = re.sub(r"\.LBB\d+_\d+", ".L", line.split(";")[0].rstrip())
            if t.strip():
                out[cur].append(t)
    return out


a, b = funcs(sys.argv[1]), funcs(sys.argv[2])
bad = 0
for k in a:
    same = a[k] == b.get(k)
    bad += not same
    print(f"{'identical' if same else ('MISSING' if k not in b else 'DIFFERENT'):10s} {len(a[k]):6d} lines  {k[:100]}")
for k in b:
    if k not in a:
        print(f"{'new':10s} {len(b[k]):6d} lines  {k[:100]}")
sys.exit(1 if bad else 0)
