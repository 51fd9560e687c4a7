#!/usr/bin/env python3
"""Exact model of the NTT backend at N = 2048 (validates the construction before it goes to HIP).  Same prime P = 2^50 - 16383
(2^14 | P - 1, so 4096-th roots of unity exist) and the same 1024-point transforms as N = 1024, arranged as in k_bootstrap_halves:
  forward : stage 0 (stride 1024, twiddle zeta_1) across the two halves, then each half a 1024-point Cooley-Tukey transform whose
            block twiddles are zeta_{k' + (1 + H) 2^floor(log2 k')} of the N = 2048 table (k' = the 1024-point transform's index)
  inverse : each half the 1024-point Gentleman-Sande transform, then the last stage (stride 1024) across the halves
  sums    : a gate's sum of 6 products can reach 2^49.58 > P/2 at N = 2048, a sum of 3 cannot (2^48.58): the b-rows and the a-rows
            are accumulated and inverse-transformed separately and added as torus words
Bounds: every intermediate below 2^53 (doubles hold exact integers), asserted on random and adversarial inputs."""
import random, sys
import numpy as np

P = (1 << 50) - 16383
PINV = 1.0 / P
N = 2048
LOGN = 11
H1 = 1024                      # sub-transform size

def is_generator(g):
    for q in (2, 3, 5, 7, 13, 19, 37, 73, 109):
        if pow(g, (P - 1) // q, P) == 1:
            return False
    return True
g = next(x for x in range(2, 100) if is_generator(x))
PSI = pow(g, (P - 1) // (2 * N), P)
assert pow(PSI, N, P) == P - 1
def brv(x, bits): return int(format(x, "0%db" % bits)[::-1], 2)
def center(x):
    x %= P
    return x - P if x > P // 2 else x
ZETA = [0] * N
for k in range(1, N):
    ZETA[k] = center(pow(PSI, brv(k, LOGN), P))
ZETA_INV = [0] + [center(pow(z, P - 2, P)) for z in ZETA[1:]]
NINV = pow(N, P - 2, P)
def sub_table(tab, H):         # table of the 1024-point sub-transform of half H, indexed by its own k' = nb' + block'
    t = [0] * H1
    for kp in range(1, H1):
        t[kp] = tab[kp + (1 + H) * (1 << (kp.bit_length() - 1))]
    return t

FWD_NORM = (5, 10)             # the 1024-point transform's own schedule (scripts/ntt/model.py)
INV_NORM = (3, 6, 9, 10)
stats = {"max_abs": 0}
def track(v):
    a = abs(v)
    if a > stats["max_abs"]: stats["max_abs"] = a
    assert a < (1 << 53), "value no longer an exact double"
    return v
def modmul(a, w):
    h = float(a) * float(w)
    q = int(np.rint(np.float64(h) * np.float64(PINV)))
    r = a * w - q * P
    assert abs(r) < 2.2 * P
    return track(r)
def normalize(x):
    q = int(np.rint(np.float64(float(x)) * np.float64(PINV)))
    return track(x - q * P)

def sub_forward(a, zt):        # 1024 points, Cooley-Tukey, natural in -> bit-reversed out
    a = list(a); h = H1 // 2; stage = 0
    while h >= 1:
        nb = H1 // (2 * h)
        for b in range(nb):
            z = zt[nb + b]
            for j in range(b * 2 * h, b * 2 * h + h):
                t = modmul(a[j + h], z)
                a[j + h] = track(a[j] - t); a[j] = track(a[j] + t)
        stage += 1
        if stage in FWD_NORM: a = [normalize(x) for x in a]
        h //= 2
    return a
def sub_inverse(a, zt):
    a = [normalize(x) for x in a]; h = 1; stage = 0
    while h <= H1 // 2:
        nb = H1 // (2 * h)
        for b in range(nb):
            z = zt[nb + b]
            for j in range(b * 2 * h, b * 2 * h + h):
                u, v = a[j], a[j + h]
                a[j] = track(u + v); a[j + h] = modmul(track(u - v), z)
        stage += 1
        if stage in INV_NORM: a = [normalize(x) for x in a]
        h *= 2
    return a

def forward(a):                # small inputs (digits) or key words |x| < 2^31
    x0, x1 = a[:H1], a[H1:]
    t = [modmul(v, ZETA[1]) for v in x1]
    lo = [track(u + w) for u, w in zip(x0, t)]
    hi = [track(u - w) for u, w in zip(x0, t)]
    return sub_forward(lo, sub_table(ZETA, 0)), sub_forward(hi, sub_table(ZETA, 1))
def inverse(lo, hi):
    u = sub_inverse(lo, sub_table(ZETA_INV, 0)); v = sub_inverse(hi, sub_table(ZETA_INV, 1))
    out0 = [normalize(track(p + q)) for p, q in zip(u, v)]
    out1 = [normalize(modmul(track(p - q), ZETA_INV[1])) for p, q in zip(u, v)]
    return out0 + out1

def negacyclic(a, b):
    full = np.convolve(np.array(a, dtype=object), np.array(b, dtype=object))
    r = [int(x) for x in full[:N]]
    for k in range(N, 2 * N - 1): r[k - N] -= int(full[k])
    return r

def three_row_product(rows, digs):
    acc = ([0] * H1, [0] * H1)
    for r_, d_ in zip(rows, digs):
        fr = [[center(x * NINV) for x in half] for half in forward(r_)]
        fd = forward(d_)
        for H in range(2):
            for k in range(H1): acc[H][k] = track(acc[H][k] + modmul(fd[H][k], fr[H][k]))
    return inverse(acc[0], acc[1])

if __name__ == "__main__":
    random.seed(2)
    a = [random.randrange(-32, 32) for _ in range(N)]
    lo, hi = forward(a)
    # the two halves together are the 2048-point transform: point p of half H is the evaluation at psi^(2 brv(H*1024 + p) + 1)
    for (H, p) in ((0, 0), (0, 5), (1, 0), (1, 1023), (1, 77)):
        root = pow(PSI, 2 * brv(H * H1 + p, LOGN) + 1, P)
        val = sum(c * pow(root, i, P) for i, c in enumerate(a)) % P
        assert ((hi if H else lo)[p] - val) % P == 0, (H, p)
    rows = [[random.randrange(-2 ** 31, 2 ** 31) for _ in range(N)] for _ in range(3)]
    digs = [[random.randrange(-32, 32) for _ in range(N)] for _ in range(3)]
    out = three_row_product(rows, digs)
    exact = [0] * N
    for r_, d_ in zip(rows, digs): exact = [x + y for x, y in zip(exact, negacyclic(d_, r_))]
    assert max(abs(x) for x in exact) < P // 2
    assert out == exact, "NTT product != exact negacyclic product"
    for dval, rval in ((-32, -2 ** 31), (31, 2 ** 31 - 1)):
        for alt in (False, True):
            rows = [[rval * (-1 if (alt and i % 2) else 1) for i in range(N)] for _ in range(3)]
            digs = [[dval * (-1 if (alt and (i // 3) % 2) else 1) for i in range(N)] for _ in range(3)]
            out = three_row_product(rows, digs)
            e1 = negacyclic(digs[0], rows[0])
            assert max(abs(3 * x) for x in e1) < P // 2
            assert out == [3 * x for x in e1]
    print("ok: generator", g, "psi", PSI, "max |value| = 2^%.2f" % np.log2(float(stats["max_abs"])),
          "worst 3-row sum 2^%.2f vs P/2 = 2^%.2f" % (np.log2(3.0 * N * 32 * 2 ** 31), np.log2(P / 2)))
