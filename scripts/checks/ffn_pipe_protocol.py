"""Brute-force check of the LDS ring protocol of ffn_fwd_pipe_kernel (csrc/ffn_fused.hip) - no GPU involved: the kernel's
program order per wave class is written down as a list of events between barriers, and for every chunk count n >= 2 the
script verifies, for every fragment read of W1(c) / W2(c) (incl. the 4-fragment run-ahead of the A ring into the next MFMA
stage), that
  (RAW) the DMA of that half-chunk is guaranteed to have landed: its movers waited for it (counted wait, loads return in
        order) BEFORE a barrier that the reader passed before the read, and
  (WAR) the next DMA into the same slot half is issued by its movers AFTER a barrier that the reader reaches after the read.
Wave classes: E = waves 0-3 (barrier in front of G1; they move the W1 halves, lead 3), L = waves 4-7 (barrier in front of X
unless `stagger` is off; they move the W2 halves, lead 2).  Barrier k = sync(k); barrier -1 = the one in front of the loop."""
import sys

NBUF = 4


def program(cls, n, stagger):
    """-> list of ('bar', k) / ('read', half, chunk) in program order; reads between two barriers are unordered"""
    late = (cls == "L") and stagger
    ev = [("bar", -1)]
    ev += [("read", "W1", 0)]                       # ring primed with W1(0)[0..3]

    def g1(k, cont):
        return [("read", "W1", k), ("read",) + cont]

    def x(k, cont):
        return [("read", "W2", k - 1), ("read",) + cont]
    # chunk 0
    if not late:
        ev.append(("bar", 0))
    ev += g1(0, ("W1", 1))
    if late:
        ev.append(("bar", 0))
    for k in range(1, n):
        if not late:
            ev.append(("bar", k))
        ev += g1(k, ("W2", k - 1))
        if late:
            ev.append(("bar", k))
        ev += x(k, ("W1", k + 1) if k + 1 < n else ("W2", k))
    ev += [("read", "W2", n - 1)]                    # final G2
    return ev


def issued_at(half, c):
    """barrier index behind which the movers issue the DMA of (half, c): prologue = -2 (before barrier -1)"""
    lag = 0 if half == "W1" else 1
    # prologue: c + lag < 3;  sync(k) issues chunk k + 3 - lag
    return -2 if c + lag < 3 else c - 3 + lag


def landed_after(half, c, n):
    """first barrier after which (half, c) is guaranteed readable: the movers' wait in front of barrier k covers every DMA
    but the youngest one when k + 2 < n (vmcnt(4)), everything otherwise (vmcnt(0)); barrier -1: vmcnt(4) if n > 2"""
    lag = 0 if half == "W1" else 1
    for k in range(-1, n):
        # youngest chunk issued by the movers before the wait in front of barrier k
        youngest = min(n - 1, (2 - lag) if k == -1 else (k - 1) + 3 - lag)
        if k == -1:
            youngest = min(n - 1, 2 - lag)
            counted = n > 2
        else:
            youngest = min(n - 1, max(2 - lag, k - 1 + 3 - lag))
            counted = k + 2 < n
        covered = youngest - 1 if counted else youngest
        if c <= covered:
            return k
    return None


def check(n, stagger):
    bad = []
    for cls in ("E", "L"):
        ev = program(cls, n, stagger)
        last_bar = None
        for i, e in enumerate(ev):
            if e[0] == "bar":
                last_bar = e[1]
                continue
            _, half, c = e
            if c >= n or c < 0:
                bad.append((cls, e, "reads a chunk that does not exist"))
                continue
            la = landed_after(half, c, n)
            if la is None or last_bar is None or la > last_bar:
                bad.append((cls, e, f"RAW: readable after barrier {la}, read after barrier {last_bar}"))
            # next DMA into the same slot half: chunk c + NBUF
            nxt = c + NBUF
            if nxt < n:
                ib = issued_at(half, nxt)
                next_bar = next((f[1] for f in ev[i + 1:] if f[0] == "bar"), None)
                # the overwrite is issued behind barrier ib; the reader must arrive at barrier ib after this read
                if next_bar is None or next_bar > ib:
                    bad.append((cls, e, f"WAR: overwritten behind barrier {ib}, reader's next barrier {next_bar}"))
    return bad


def main():
    total = 0
    for stagger in (True, False):
        for n in range(2, 33):
            bad = check(n, stagger)
            total += len(bad)
            for b in bad[:6]:
                print(f"n = {n:2d} stagger = {stagger}: {b}")
    print("violations:", total)
    sys.exit(1 if total else 0)


if __name__ == "__main__":
    main()
