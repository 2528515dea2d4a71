"""Check of the counted waits of attn_block_fwd_kernel<NBUF = 4> (csrc/attn_fused.hip), no GPU involved.  Per wave, the
vector-memory operations are listed in program order (DMA pieces of chunk c, training stores, residual-row loads, row
stores); at every sync(k) the wait `s_waitcnt vmcnt(P)` leaves at most P operations outstanding.  Loads return in order
(stores may complete in any order), so a load is certainly complete iff more than P LOADS sit at or behind it in program
order... the script verifies that every piece of chunk k + 1 is certainly complete at sync(k) - for early waves (sync in
front of a stage) and late waves (behind it), with and without training stores, with 4 or 8 residual loads per out_proj
iteration - and the slot-reuse order (chunk k + 3 overwrites chunk k - 1 behind barrier k)."""
import itertools
import sys

N_CHUNK = 20


def pieces(c):
    return 2 if (c < 16 and (c & 1)) else 4


def wait_value(k):
    nx = k + 2
    return 2 if (nx < 16 and (nx & 1)) else 4


def ops_of_wave(late, train_stores, res_loads):
    """-> program-ordered list of events: ('dma', c) per piece, ('st',), ('ld',), ('sync', k)"""
    ev = []
    for c in (0, 1, 2):
        ev += [("dma", c)] * pieces(c)
    ev.append(("drain",))                      # vmcnt(0) + barrier in front of the loop

    def sync(k):
        out = [("sync", k)]
        if k + 3 < N_CHUNK:
            out += [("dma", k + 3)] * pieces(k + 3)
        done = k if late else k - 1
        if train_stores and 0 <= done < 16:
            out += [("st",)] * train_stores
        return out
    for h in range(8):
        if not late:
            ev += sync(2 * h)
        # stage A (no vector-memory operations of its own)
        ev += sync(2 * h) if late else sync(2 * h + 1)
        # stage B
        if late:
            ev += sync(2 * h + 1)
    for u in range(4):
        if not late:
            ev += sync(16 + u)
        ev += [("ld",)] * res_loads            # residual rows (+ the per-sequence rows)
        ev += [("st",)] * 4                    # the rows of both tiles, behind their arithmetic
        if late:
            ev += sync(16 + u)
    return ev


def check():
    bad = 0
    for late, ts, rl in itertools.product((False, True), (0, 2, 4), (4, 8)):
        ev = ops_of_wave(late, ts, rl)
        issued = []                            # ('dma', c) / ('ld',) / ('st',) since the last full drain
        for e in ev:
            if e[0] == "drain":
                issued = []
            elif e[0] == "sync":
                k = e[1]
                P = wait_value(k)
                # loads certainly complete: those with more than P loads at or behind them
                loads = [x for x in issued if x[0] in ("dma", "ld")]
                need = [i for i, x in enumerate(loads) if x[0] == "dma" and x[1] <= k + 1]
                for i in need:
                    if len(loads) - i <= P:
                        print(f"late={late} stores={ts} res={rl}: sync({k}) vmcnt({P}) does not cover a piece of chunk {loads[i][1]}")
                        bad += 1
                # (operations certainly complete need not be tracked further; keep the list: the bound is conservative)
            else:
                issued.append(e)
    # slot reuse: chunk k + 3 goes into slot (k + 3) % 4 = (k - 1) % 4 behind barrier k; readers of chunk k - 1: early waves'
    # stage k - 1 ends before their sync(k), late waves' stage k - 1 ends before their sync(k - 1) - always in front of barrier k
    return bad


if __name__ == "__main__":
    b = check()
    print("violations:", b)
    sys.exit(1 if b else 0)
