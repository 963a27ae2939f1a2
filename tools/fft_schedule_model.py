#!/usr/bin/env python3
"""CPU model of the register-resident FFT schedule of srla_autocorr_w (kernels.hip): checks, in exact IEEE doubles and
without a GPU, that the schedule computes THE SAME butterflies on THE SAME operands as the reference's Stockham transform
(fft.c:71-136 as restated in oracle/srla_oracle.c), i.e. that its results are bit-identical, and that every index formula
the kernel uses (register <-> slot maps of the three passes, twiddle indices, the digit-reversed place of every output,
the in-lane pairing of bins i and m - i, the pruning of the inverse) is right.

    python tools/fft_schedule_model.py            # all supported sizes

The schedule (T lanes per transform of m = 32 T complex points, 32 complex values per lane, in place):
  pass 1  stages 1-2 in registers; lane t holds the residue classes cA, cB modulo n2 = m / 16: slot = class + n2 j, j < 16
  T1      through LDS: slot order
  pass 2  stages 3-4; lane t holds units u = t, t + T; unit (B, c) = slots B n2 + 4 C a + C b + c (C = m / 256; a, b < 4)
  T2      through LDS
  pass 3  the remaining stages (C points each: radix 2 / 4 / 4+2 / 4+4); lane t holds units v = t + T g of C contiguous slots
  T3      X[k] sits at slot rev(k); the spectrum pass wants bins i and m - i in one lane
"""
import math
import random
import sys


def c_add(a, b): return (a[0] + b[0], a[1] + b[1])
def c_sub(a, b): return (a[0] - b[0], a[1] - b[1])
def c_mul(a, b): return (a[0] * b[0] - a[1] * b[1], a[0] * b[1] + a[1] * b[0])


def tables(m, flag):
    """per stage (sub-size n > 2): (w1[p], w2[p], w3[p]) for p < n / 4, by the reference's recurrences (fft.c:83-107)"""
    out = []
    n = m
    while n > 2:
        theta = 2.0 * math.pi / n
        step = (math.cos(theta), flag * math.sin(theta))
        w1 = (1.0, 0.0)
        t1, t2, t3 = [], [], []
        for p in range(n >> 2):
            w2 = c_mul(w1, w1)
            w3 = c_mul(w1, w2)
            t1.append(w1); t2.append(w2); t3.append(w3)
            w1 = c_mul(w1, step)
        out.append((t1, t2, t3))
        n >>= 2
    return out


def butterfly(a, b, c, d, flag, w1, w2, w3):
    apc, amc, bpd, bmd = c_add(a, c), c_sub(a, c), c_add(b, d), c_sub(b, d)
    rot = (0.0, -flag)
    jbmd = c_mul(rot, bmd)
    return (c_add(apc, bpd), c_mul(w1, c_sub(amc, jbmd)), c_mul(w2, c_sub(apc, bpd)), c_mul(w3, c_add(amc, jbmd)))


def stockham(x, flag):
    m = len(x)
    tw = tables(m, flag)
    x = list(x)
    y = [None] * m
    n, s, st = m, 1, 0
    while n > 2:
        q4, h, t4 = n >> 2, n >> 1, (n >> 2) + (n >> 1)
        t1, t2, t3 = tw[st]
        for p in range(q4):
            for q in range(s):
                o = butterfly(x[q + s * p], x[q + s * (p + q4)], x[q + s * (p + h)], x[q + s * (p + t4)], flag, t1[p], t2[p], t3[p])
                for k in range(4):
                    y[q + s * (4 * p + k)] = o[k]
        n >>= 2; s <<= 2; st += 1
        x, y = y, x
    if n == 2:
        for q in range(s):
            a, b = x[q], x[q + s]
            y[q], y[q + s] = c_add(a, b), c_sub(a, b)
        x, y = y, x
    return x


def rev4(v):
    """base-4 digit reversal of an 8-bit unit index"""
    return ((v >> 6) & 3) | (((v >> 4) & 3) << 2) | (((v >> 2) & 3) << 4) | ((v & 3) << 6)


def wrev(c, C):
    if C <= 4: return c
    if C == 8: return (c >> 1) + 4 * (c & 1)
    return (c >> 2) + 4 * (c & 3)


def classes(t, T, paired):
    n2 = 2 * T
    if not paired: return (t, t + T)
    return (0, T) if t == 0 else (t, n2 - t)


def schedule(x, flag, T, paired, need=None):
    """the kernel's schedule; returns {k: X[k]} for the outputs it produces (all, or k < need)"""
    m = len(x)
    assert m == 32 * T
    n2, C = m // 16, m // 256
    tw = tables(m, flag)
    prune = need is not None
    need = m if need is None else need
    lds = [None] * m
    # ---- pass 1: lane t, reg 16 X + j  <->  slot class_X + n2 j
    for t in range(T):
        regs = {}
        cl = classes(t, T, paired)
        for X in range(2):
            for j in range(16):
                regs[(X, j)] = x[cl[X] + n2 * j]
        for X in range(2):
            t1, t2, t3 = tw[0]
            for j0 in range(4):
                p = cl[X] + n2 * j0
                o = butterfly(regs[(X, j0)], regs[(X, j0 + 4)], regs[(X, j0 + 8)], regs[(X, j0 + 12)], flag, t1[p], t2[p], t3[p])
                for k in range(4): regs[(X, j0 + 4 * k)] = o[k]
            t1, t2, t3 = tw[1]
            for k1 in range(4):
                p = cl[X]
                o = butterfly(*[regs[(X, 4 * k1 + k)] for k in range(4)], flag, t1[p], t2[p], t3[p])
                for k in range(4): regs[(X, 4 * k1 + k)] = o[k]
        for X in range(2):
            for j in range(16):
                lds[j * n2 + cl[X]] = regs[(X, j)]          # T1 write: block B = j, position = class
    # ---- pass 2: lane t, unit h: u = t + T h = B C + c; reg 16 h + 4 a + b  <->  slot B n2 + 4 C a + C b + c
    lds2 = [None] * m
    for t in range(T):
        for h in range(2):
            u = t + T * h
            B, c = u // C, u % C
            P = (B >> 2) | ((B & 3) << 2)                  # k1 + 4 k2 (B = 4 k1 + k2)
            if prune and P >= need: continue
            r = {(a, b): lds[B * n2 + 4 * C * a + C * b + c] for a in range(4) for b in range(4)}
            t1, t2, t3 = tw[2]
            for b in range(4):
                p = C * b + c
                o = butterfly(r[(0, b)], r[(1, b)], r[(2, b)], r[(3, b)], flag, t1[p], t2[p], t3[p])
                for a in range(4): r[(a, b)] = o[a]
            for a in range(4):
                if prune and P + 16 * a >= need:
                    for b in range(4): r[(a, b)] = None
                    continue
                if C >= 2 and len(tw) > 3:
                    t1, t2, t3 = tw[3]
                    o = butterfly(r[(a, 0)], r[(a, 1)], r[(a, 2)], r[(a, 3)], flag, t1[c], t2[c], t3[c])
                    for b in range(4): r[(a, b)] = o[b] if (not prune or P + 16 * a + 64 * b < need) else None
            for a in range(4):
                for b in range(4):
                    lds2[B * n2 + 4 * C * a + C * b + c] = r[(a, b)]
    # ---- pass 3: lane t, unit g: v = t + T g; reg g C + c  <->  slot v C + c
    out = {}
    for t in range(T):
        for g in range(256 // T):
            v = t + T * g
            if prune and rev4(v) >= need: continue
            r = [lds2[v * C + c] for c in range(C)]
            if C == 2:
                r = [c_add(r[0], r[1]), c_sub(r[0], r[1])]
            elif C == 4:
                t1, t2, t3 = tw[4]
                r = list(butterfly(r[0], r[1], r[2], r[3], flag, t1[0], t2[0], t3[0]))
            elif C == 8:
                t1, t2, t3 = tw[4]
                for p in range(2):
                    o = butterfly(r[p], r[p + 2], r[p + 4], r[p + 6], flag, t1[p], t2[p], t3[p])
                    for k in range(4): r[p + 2 * k] = o[k]
                for k in range(4):
                    a, b = r[2 * k], r[2 * k + 1]
                    r[2 * k], r[2 * k + 1] = c_add(a, b), c_sub(a, b)
            else:
                t1, t2, t3 = tw[4]
                for p in range(4):
                    o = butterfly(r[p], r[p + 4], r[p + 8], r[p + 12], flag, t1[p], t2[p], t3[p])
                    for k in range(4): r[p + 4 * k] = o[k]
                t1, t2, t3 = tw[5]
                for k in range(4):
                    o = butterfly(r[4 * k], r[4 * k + 1], r[4 * k + 2], r[4 * k + 3], flag, t1[0], t2[0], t3[0])
                    for i in range(4): r[4 * k + i] = o[i]
            for c in range(C):
                k = rev4(v) + 256 * wrev(c, C)
                if k < need:
                    assert r[c] is not None, (v, c, k)
                    out[k] = r[c]
    return out


def check_pairing(T):
    """bins i and m - i of every pair live in one lane of the paired layout; every pair 1 <= i <= m / 2 exactly once"""
    m, n2 = 32 * T, 2 * T
    seen = set()
    for t in range(T):
        cA, cB = classes(t, T, True)
        mine = {cA + n2 * j for j in range(16)} | {cB + n2 * j for j in range(16)}
        for i in mine:
            if i == 0: continue
            assert (m - i) in mine, (T, t, i)
            seen.add(min(i, m - i))
    assert seen == set(range(1, m // 2 + 1))


def main():
    random.seed(5)
    bad = 0
    for T in (16, 32, 64, 128):
        m = 32 * T
        check_pairing(T)
        x = [(random.uniform(-1, 1), random.uniform(-1, 1)) for _ in range(m)]
        for flag in (-1, 1):
            want = stockham(x, flag)
            for paired in (False, True):
                got = schedule(x, flag, T, paired)
                ok = len(got) == m and all(got[k] == want[k] for k in range(m))
                print("m=%4d flag=%+d paired=%d full: %s" % (m, flag, paired, "identical" if ok else "DIFFERENT"))
                bad += not ok
        for need in (9, 17, 33, 132):
            got = schedule(x, 1, T, True, need)
            want = stockham(x, 1)
            ok = all(k in got and got[k] == want[k] for k in range(need))
            print("m=%4d inverse pruned to %3d: %s" % (m, need, "identical" if ok else "DIFFERENT"))
            bad += not ok
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
