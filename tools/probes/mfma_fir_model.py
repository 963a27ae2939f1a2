"""numpy model of residual_cost.hip: mfma_fir -- the FIR as tiles of a 16 x 16 x 64 int8 product on signed byte digits, the lane
permutation that returns every chunk to its owner; checked against the direct sum for FL = 1..8 and orders 1..255
(python tools/probes/mfma_fir_model.py; tests/test_kernel_models.py runs a subset).  MF_OFFZ / MF_TZB are residual_cost.hip's."""
import numpy as np
rng = np.random.default_rng(1)
MF_OFFZ, MF_TZB = 160, 576
def emul(FL, order, n=None):
    S = 4*FL; n = 1024*FL
    x = rng.integers(-40000, 40000, size=n).astype(np.int64)      # beyond 16 bits: exercises the third plane
    coef = rng.integers(-128, 128, size=order).astype(np.int64)
    half = 1 << 6
    # reference: acc[t] = half + sum_k coef[k] * x[t - order + k]  (x[<0] = 0)
    xp = np.concatenate([np.zeros(512, np.int64), x])   # (512 >= the largest order)
    ref = np.array([half + sum(coef[k]*xp[512 + t - order + k] for k in range(order)) for t in range(n)], dtype=np.int64) & 0xFFFFFFFF
    # planes (signed digits)
    s0 = ((x + 128) & 0xFF) - 128; x1 = (x - s0) >> 8; s1 = ((x1 + 128) & 0xFF) - 128; x2 = (x1 - s1) >> 8
    assert np.all(x == s0 + 256*s1 + 65536*x2) and np.all(np.abs(x2) < 128)
    PADB = 256
    planes = [np.concatenate([np.zeros(PADB, np.int64), p, rng.integers(-128,128,size=1024)]) for p in (s0, s1, x2)]
    P2 = (order + 15) & ~15; D = P2 - order
    NKB = (16*FL - 1 + P2 + 63)//64
    OFFZ = MF_OFFZ
    assert OFFZ + 64*NKB + 16 <= MF_TZB, (FL, order)
    tapz = np.zeros(MF_TZB, np.int64); tapz[OFFZ:OFFZ+order] = coef
    out = np.zeros(n, np.int64)
    for wave in range(4):
        wvbase = 64*S*wave
        accs = np.zeros((FL, 64, 4), np.int64)     # [tile][lane][i]
        for T in range(FL):
            acc = np.zeros((16, 16), np.int64)     # [row][col]
            for kb in range(NKB):
                A = np.zeros((16, 64), np.int64); B = np.zeros((64, 16), np.int64)
                for lane in range(64):
                    rr, gk = lane & 15, lane >> 4
                    base = 16*gk - 4*FL*(rr >> 2) - (rr & 3) - D
                    assert OFFZ + base + 64*kb - 4*T >= 0 and OFFZ + base + 64*kb - 4*T + 16 <= MF_TZB
                    for t in range(16):
                        A[rr, 16*gk + t] = tapz[OFFZ + base + 64*kb - 4*T + t]
                    cc = lane & 15
                    b0 = PADB + wvbase + 16*FL*cc - P2 + 64*kb + 16*gk
                    for t in range(16):
                        B[16*gk + t, cc] = 0   # filled per plane below
                # planes combined: do per plane products
                tot = np.zeros((16,16), np.int64)
                for pi, sh in ((0,0),(1,8),(2,16)):
                    for lane in range(64):
                        cc, gk = lane & 15, lane >> 4
                        b0 = PADB + wvbase + 16*FL*cc - P2 + 64*kb + 16*gk
                        assert b0 % 16 == 0 and b0 >= 0
                        B[16*gk:16*gk+16, cc] = planes[pi][b0:b0+16]
                    tot += (A @ B) << sh
                acc += tot
            for lane in range(64):
                cc, g = lane & 15, lane >> 4
                for i in range(4):
                    accs[T, lane, i] = acc[4*g + i, cc]
        # bpermute: owner lane l' pulls from src = 16*(l'&3) + (l'>>2)
        for lp in range(64):
            src = 16*(lp & 3) + (lp >> 2)
            for T in range(FL):
                for i in range(4):
                    out[wvbase + S*lp + 4*T + i] = (half + accs[T, src, i]) & 0xFFFFFFFF
    return np.array_equal(out, ref)
if __name__ == "__main__":
    for FL in range(1, 9):
        for order in (1, 5, 16, 17, 32, 33, 48, 55, 64, 128, 255):
            print(FL, order, emul(FL, order))
