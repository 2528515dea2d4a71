"""Index-level check (numpy, no GPU) that the sliced E1 of ffn_fwd_pipe_kernel produces the same packed hidden tile as
ffn_fwd_kernel's E1: both per-lane programs are transcribed from csrc/ffn_fused.hip (bias quad addressing, ReLU, the two draw
words per quad, mask / scale, the order of the packed words in hf[0] / hf[1]) and run on random accumulators for every lane
half, chunk and a set of rows, with the 64-bit group index of the original against the (gl, gh) form of the variant."""
import numpy as np

M32 = 0xFFFFFFFF
CH, FF = 32, 512


def hash32(x):
    x &= M32
    x ^= x >> 16; x = (x * 0x7feb352d) & M32
    x ^= x >> 15; x = (x * 0x846ca68b) & M32
    x ^= x >> 16
    return x


def drop2_word(h, i):
    w = (h + (i + 1) * 0x9e3779b9) & M32
    w ^= w >> 16; w = (w * 0x7feb352d) & M32; w ^= w >> 15
    return w


def drop2_group(s0, s1, g16):
    h = hash32((g16 & M32) ^ s0)
    return ((h ^ s1) + ((g16 >> 32) & M32) * 0x9e3779b1) & M32


def bf16_pair(lo, hi):
    """two fp32 -> packed bf16 word (round to nearest even), lo in bits 0..15"""
    def cv(f):
        u = int(np.float32(f).view(np.uint32))
        return ((u + 0x7fff + ((u >> 16) & 1)) >> 16) & 0xffff
    return cv(lo) | (cv(hi) << 16)


def e1_original(hid, b1, row, c, half, s0, s1, thresh, scale, drop):
    """ffn_fwd_kernel::E1 -> (hf[0] words, hf[1] words)"""
    id0 = row * FF + CH * c + 16 * half
    hh = drop2_group(s0, s1, id0 >> 4) if drop else 0
    out = []
    for ks2 in range(2):
        v = [0.0] * 8
        for qq in range(2):
            q = 2 * ks2 + qq
            bb = b1[CH * c + 8 * q + 4 * half: CH * c + 8 * q + 4 * half + 4]
            for e in range(4):
                v[4 * qq + e] = max(np.float32(hid[4 * q + e] + bb[e]), np.float32(0))
        if drop:
            m = [0.0] * 8
            for i in range(4):
                w = drop2_word(hh, 4 * ks2 + i)
                m[2 * i] = 0.0 if (w & 0xffff) < thresh else scale
                m[2 * i + 1] = 0.0 if (w >> 16) < thresh else scale
            v = [np.float32(a * np.float32(b)) for a, b in zip(v, m)]
        out.append([bf16_pair(v[0], v[1]), bf16_pair(v[2], v[3]), bf16_pair(v[4], v[5]), bf16_pair(v[6], v[7])])
    return out


def e1_sliced(hid, b1, row, c, half, s0, s1, thresh, scale, drop):
    """ffn_fwd_pipe_kernel::X, the 12 slices + the two pack points"""
    hid = [np.float32(x) for x in hid]
    gl = ((row << 5) & M32) | half
    gh = ((row >> 27) * 0x9e3779b1) & M32
    hh = wd0 = wd1 = 0
    bq = b1[CH * c + 4 * half: CH * c + 4 * half + 4]            # fetched under G1's tail: quad 0
    hf = [None, None]
    for n in range(16):
        if n < 12:
            q, ph = n // 3, n % 3
            if ph == 0:
                if drop and q == 0:
                    hh = ((hash32((gl | (2 * c)) ^ s0) ^ s1) + gh) & M32
                for e in range(4):
                    hid[4 * q + e] = max(np.float32(hid[4 * q + e] + bq[e]), np.float32(0))
            elif ph == 1:
                if drop:
                    wd0, wd1 = drop2_word(hh, 2 * q), drop2_word(hh, 2 * q + 1)
            else:
                if drop:
                    mm = [0.0 if (wd0 & 0xffff) < thresh else scale, 0.0 if (wd0 >> 16) < thresh else scale,
                          0.0 if (wd1 & 0xffff) < thresh else scale, 0.0 if (wd1 >> 16) < thresh else scale]
                    for e in range(4):
                        hid[4 * q + e] = np.float32(hid[4 * q + e] * np.float32(mm[e]))
                if q < 3:
                    o = CH * c + 8 * (q + 1) + 4 * half
                    bq = b1[o:o + 4]
        if n == 8:
            hf[0] = [bf16_pair(hid[0], hid[1]), bf16_pair(hid[2], hid[3]), bf16_pair(hid[4], hid[5]), bf16_pair(hid[6], hid[7])]
    hf[1] = [bf16_pair(hid[8], hid[9]), bf16_pair(hid[10], hid[11]), bf16_pair(hid[12], hid[13]), bf16_pair(hid[14], hid[15])]
    return hf


def main():
    rng = np.random.default_rng(0)
    b1 = rng.standard_normal(FF).astype(np.float32)
    s0, s1 = 0x12345678, 0x9abcdef1
    thresh, scale = 6554, np.float32(65536.0 / (65536 - 6554))
    n = bad = 0
    for row in (0, 1, 31, 4097, 126975, (1 << 27) + 5, (1 << 31) - 300):
        for c in range(16):
            for half in (0, 1):
                for drop in (False, True):
                    hid = rng.standard_normal(16).astype(np.float32) * 2
                    a = e1_original(hid, b1, row, c, half, s0, s1, thresh, scale, drop)
                    b = e1_sliced(hid, b1, row, c, half, s0, s1, thresh, scale, drop)
                    n += 1
                    if a != b:
                        bad += 1
                        if bad < 5:
                            print("MISMATCH", row, c, half, drop, a, b)
    print(f"{n} cases, {bad} mismatches")
    raise SystemExit(1 if bad else 0)


if __name__ == "__main__":
    main()
