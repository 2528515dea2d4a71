"""Static instruction mix of the MFMA loops of every kernel: per loop (LLVM's loop-header comments in the `hipcc -S` output)
the number of MFMA, VALU, LDS, VMEM, SALU, s_waitcnt and s_nop instructions and instructions per MFMA; with --spills also
every kernel that has scratch (spill) instructions and how many of them sit inside a loop.
Why it matters on gfx950: a wave issues one instruction every ~5 cycles whatever the opcode (scripts/probes/
valu_rate_probe.hip), a 32x32x16 bf16 MFMA occupies the SIMD's matrix pipe for 32 cycles; with w waves per SIMD a loop
whose instructions-per-MFMA exceed 6.4 w is bound by instruction issue, not by the matrix pipe.
usage: bash scripts/isa_all.sh /tmp/isa && python scripts/isa_loop_mix.py [--spills] /tmp/isa/*.s"""
import collections
import re
import sys


def kernels(path):
    lines = open(path).read().split("\n")
    starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
    ends = [i for i, l in enumerate(lines) if l.startswith(".Lfunc_end")]
    for (i, name), e in zip(starts, ends):
        yield name, lines[i:e]


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "scratch_", "flat_")):
        return "vmem"
    return "other"


def main():
    spills = "--spills" in sys.argv
    for path in [a for a in sys.argv[1:] if not a.startswith("--")]:
        for name, body in kernels(path):
            cur, loops = None, collections.OrderedDict()
            scratch = in_loop = 0
            for l in body:
                m = re.match(r"^(\.LBB\d+_\d+):", l)
                if m or l.startswith("; %bb."):
                    hm = re.search(r"Header=(BB\d+_\d+)", l)
                    cur = m.group(1)[2:] if (m and "Loop Header" in l) else (hm.group(1) if hm else None)
                    continue
                t = l.split(";")[0].strip()
                if not t or t.startswith(".") or t.endswith(":"):
                    continue
                if "scratch_" in t:
                    scratch += 1
                    in_loop += cur is not None
                if cur:
                    d = loops.setdefault(cur, collections.Counter())
                    d["total"] += 1
                    d[classify(t.split()[0])] += 1
            short = re.sub(r"^_ZN12_GLOBAL__N_1\d+|^_Z\d+", "", name)[:70]
            for h, d in loops.items():
                if d["mfma"] >= 8:
                    print(f"{short:70s} loop {h:9s} mfma {d['mfma']:3d} valu {d['valu']:4d} lds {d['lds']:3d} vmem {d['vmem']:3d} "
                          f"salu {d['salu']:3d} wait {d['wait']:3d} nop {d['nop']:3d} | per mfma {d['total'] / d['mfma']:5.1f}")
            if spills and scratch:
                print(f"{short:70s} scratch instructions {scratch:3d}, inside loops {in_loop:3d}")


if __name__ == "__main__":
    main()
