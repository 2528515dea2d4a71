"""Per-kernel summary of a hipcc --save-temps .s file: line count, MFMA count, scratch (spill) instructions and where they
sit relative to the MFMA sequence, s_waitcnt vmcnt(0) count.  usage: python scripts/isa_scan.py file.s [name filter]"""
import bisect
import re
import sys
from collections import Counter

lines = open(sys.argv[1]).read().split("\n")
flt = sys.argv[2] if len(sys.argv) > 2 else ""
starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
ends = [i for i, l in enumerate(lines) if l.startswith(".Lfunc_end")]
for (i, name), e in zip(starts, ends):
    if flt not in name:
        continue
    body = lines[i:e]
    sc = [j for j, l in enumerate(body) if "scratch_" in l]
    mf = [j for j, l in enumerate(body) if "v_mfma" in l]
    print(name[:110])
    print(f"  lines {len(body)}  mfma {len(mf)}  scratch ops {len(sc)}  vmcnt(0) {sum('vmcnt(0)' in l for l in body)}"
          f"  ds_read {sum('ds_read' in l for l in body)}  global_load {sum('global_load' in l for l in body)}"
          f"  global_store {sum('global_store' in l for l in body)}")
    if sc:
        pos = Counter(bisect.bisect(mf, j) for j in sc)
        print("  scratch ops after MFMA #:", sorted(pos.items()))
