"""Counted waits of ffn_bwd_one_kernel (csrc/ffn_fused.hip), derived and checked without a GPU.  Per wave the vector-memory
LOADS are listed in program order - weight chunks W(c) (4 LDS-DMA pieces per wave), gate pieces h(c) (2 LDS-DMA pieces per
wave into the wave's private staging slot c & 1) - together with the two wait points of every iteration:
    sync(c):  needs W(c + 1) landed (then the barrier makes every wave's pieces of it visible)
    gate(c):  needs h(c) landed
Loads return in order, so `s_waitcnt vmcnt(N)` certainly covers a load iff more than N loads sit at or behind it in program
order at the wait (stores in flight only make the wait stricter).  The script prints, for early (sync in front of G1) and
late (sync in front of G2) waves, the largest admissible N per iteration and checks the constants the kernel uses:
    4 slots:  sync(c): vmcnt(8) while c + 2 < NCH, else vmcnt(0);   gate(c): vmcnt(2) for c < 2, vmcnt(10) while c + 3 < NCH, else 0
    3 slots:  sync(c): vmcnt(2) while c + 2 < NCH, else vmcnt(0);   gate(c): vmcnt(2) for c < 2, vmcnt(10) while c + 2 < NCH, else 0
and the reuse of the staging slots (h(c + 2) is issued behind gate(c)'s reads) and ring slots (W(c + 3) behind barrier c)."""
import sys

NCH = 16
NBUF = 4                    # ring slots: the weight DMA runs DIST = NBUF - 1 chunks ahead


def sync_wait(c):
    if NBUF == 4:
        return 8 if c + 2 < NCH else 0
    return 2 if c + 2 < NCH else 0          # 3 slots: only the two gate pieces of chunk c + 1 / c + 2 sit behind W(c + 1)


def gate_wait(c):
    if c < 2:
        return 2            # (the first two gate pieces sit right behind the prologue: few loads behind them yet)
    if NBUF == 4:
        return 10 if c + 3 < NCH else 0
    return 10 if c + 2 < NCH else 0


def program(late):
    dist = NBUF - 1
    ev = []
    for c in range(dist):
        ev += [("W", c)] * 4
    ev += [("drain_loads",)] + [("h", 0)] * 2 + [("h", 1)] * 2
    ev.append(("pre",))

    def sync(c):
        out = [("sync", c)]
        if c + dist < NCH:
            out += [("W", c + dist)] * 4
        return out
    for c in range(NCH):
        if not late:
            ev += sync(c)
        ev.append(("gate", c))
        if c + 2 < NCH:
            ev += [("h", c + 2)] * 2
        if late:
            ev += sync(c)
    return ev


def check():
    bad = 0
    for late in (False, True):
        loads = []          # program-ordered loads since the start (complete ones stay: the bound is positional)
        done_upto = 0       # loads before this index are certainly complete (the prologue's dy loads drain them)
        admissible = {"sync": [], "gate": []}
        for e in program(late):
            if e[0] in ("W", "h"):
                loads.append(e)
            elif e[0] == "drain_loads":
                done_upto = len(loads)          # the compiler's waits for the dy rows (younger loads) cover every older load
            elif e[0] in ("sync", "gate"):
                c = e[1]
                need = ("W", c + 1) if e[0] == "sync" else ("h", c)
                N = sync_wait(c) if e[0] == "sync" else gate_wait(c)
                idx = [i for i, l in enumerate(loads) if l == need and i >= done_upto]
                if need[1] >= NCH or not idx:
                    admissible[e[0]].append(None)
                    continue
                last = max(idx)
                room = len(loads) - last - 1                    # loads behind the last needed piece
                admissible[e[0]].append(room)
                if N > room:
                    print(f"late={late}: {e[0]}({c}) vmcnt({N}) does not cover {need}: only {room} loads behind it")
                    bad += 1
        print(f"late={late}: largest admissible vmcnt at sync(c): {admissible['sync']}")
        print(f"late={late}: largest admissible vmcnt at gate(c): {admissible['gate']}")
    return bad


if __name__ == "__main__":
    b = 0
    for NBUF in (4, 3):
        print(f"--- {NBUF} ring slots")
        b += check()
    print("violations:", b)
    sys.exit(1 if b else 0)
