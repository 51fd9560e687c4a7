#!/usr/bin/env python3
"""Exact model of the NTT backend (host side; validates tables, operation bounds and the algorithm before it goes to HIP).
Prime P = 2^50 - 16383, negacyclic NTT of size N = 1024 with merged twist:
  forward  = Cooley-Tukey butterflies, stride N/2 .. 1, twiddle per block  (natural in -> bit-reversed out)
  inverse  = Gentleman-Sande butterflies, stride 1 .. N/2, inverse twiddles (bit-reversed in -> natural out), N^-1 folded into the key
Doubles hold exact integers; modmul(a, w) = a*w - rint(fl(fl(a*w) * Pinv)) * P  (6 DP ops with FMA on the device)."""
import random, sys
import numpy as np

P = (1 << 50) - 16383
N = 1024
PINV = 1.0 / P

def is_generator(g):
    # P - 1 = 2^14 * (2^36 - 1);  2^36 - 1 = 3^3 * 5 * 7 * 13 * 19 * 37 * 73 * 109
    for q in (2, 3, 5, 7, 13, 19, 37, 73, 109):
        if pow(g, (P - 1) // q, P) == 1:
            return False
    return True

assert (P - 1) % (1 << 14) == 0 and ((P - 1) >> 14) == (1 << 36) - 1
g = next(x for x in range(2, 100) if is_generator(x))
PSI = pow(g, (P - 1) // (2 * N), P)          # primitive 2N-th root of unity
assert pow(PSI, N, P) == P - 1

def brv(x, bits):
    return int(format(x, "0%db" % bits)[::-1], 2)

def center(x):
    x %= P
    return x - P if x > P // 2 else x

LOGN = N.bit_length() - 1
ZETA = [0] * N          # ZETA[k], k = nb + block (nb = number of blocks of the stage), as in the usual merged-twist tables
for k in range(1, N):
    ZETA[k] = center(pow(PSI, brv(k, LOGN), P))
ZETA_INV = [0] + [center(pow(z, P - 2, P)) for z in ZETA[1:]]
NINV = pow(N, P - 2, P)

FWD_NORM = (5, 10)       # device: after the first stage of pass 2, and at the end
INV_NORM = (3, 6, 9, 10) # device: plus one normalisation of the input sums
stats = {"max_abs": 0}
def track(v):
    a = abs(v)
    if a > stats["max_abs"]:
        stats["max_abs"] = a
    assert a < (1 << 53), "value no longer an exact double"
    return v

def modmul(a, w):
    h = float(a) * float(w)                  # fl(a*w)
    q = int(np.rint(np.float64(h) * np.float64(PINV)))
    r = a * w - q * P                        # = fma(-q, P, h) + fma(a, w, -h), both exact (checked by the bound below)
    assert abs(r) < 2.2 * P
    return track(r)

def normalize(x):
    q = int(np.rint(np.float64(float(x)) * np.float64(PINV)))
    return track(x - q * P)

def forward(a):
    a = list(a)
    h = N // 2
    stage = 0
    while h >= 1:
        nb = N // (2 * h)
        for b in range(nb):
            z = ZETA[nb + b]
            for j in range(b * 2 * h, b * 2 * h + h):
                t = modmul(a[j + h], z)
                a[j + h] = track(a[j] - t)
                a[j] = track(a[j] + t)
        stage += 1
        if stage in FWD_NORM:                # renormalisation points (stage counts)
            a = [normalize(x) for x in a]
        h //= 2
    return a

def inverse(a):
    a = [normalize(x) for x in a]
    h = 1
    stage = 0
    while h <= N // 2:
        nb = N // (2 * h)
        for b in range(nb):
            z = ZETA_INV[nb + b]
            for j in range(b * 2 * h, b * 2 * h + h):
                u, v = a[j], a[j + h]
                a[j] = track(u + v)
                a[j + h] = modmul(track(u - v), z)
        stage += 1
        if stage in INV_NORM:                # sums double every stage
            a = [normalize(x) for x in a]
        h *= 2
    return a

def negacyclic(a, b):
    r = [0] * N
    for i in range(N):
        if a[i] == 0: continue
        for j in range(N):
            k = i + j
            if k < N: r[k] += a[i] * b[j]
            else: r[k - N] -= a[i] * b[j]
    return r

if __name__ == "__main__":
    random.seed(1)
    # forward really evaluates at odd powers of psi in bit-reversed order
    a = [random.randrange(-32, 32) for _ in range(N)]
    fa = forward(a)
    for p in (0, 1, 5, 1023):
        root = pow(PSI, 2 * brv(p, LOGN) + 1, P)
        assert (fa[p] - sum(c * pow(root, i, P) for i, c in enumerate(a))) % P == 0
    # external-product-like accumulation: 6 rows, torus x digits, N^-1 folded into the key rows
    rows = [[random.randrange(-2 ** 31, 2 ** 31) for _ in range(N)] for _ in range(6)]
    digs = [[random.randrange(-32, 32) for _ in range(N)] for _ in range(6)]
    acc = [0] * N
    for r_, d_ in zip(rows, digs):
        fr = [center(x * NINV) for x in forward(r_)]
        fd = forward(d_)
        for k in range(N):
            acc[k] = track(acc[k] + modmul(fd[k], fr[k]))
    out = inverse(acc)
    exact = [0] * N
    for r_, d_ in zip(rows, digs):
        e = negacyclic(d_, r_)
        exact = [x + y for x, y in zip(exact, e)]
    assert max(abs(x) for x in exact) < P // 2
    assert out == exact, "NTT product != exact negacyclic product"
    # adversarial magnitudes: every digit -32, every key word -2^31 (largest possible true sum), and alternating signs
    for dval, rval in ((-32, -2 ** 31), (31, 2 ** 31 - 1)):
        for alt in (False, True):
            rows = [[rval * (-1 if (alt and i % 2) else 1) for i in range(N)] for _ in range(6)]
            digs = [[dval * (-1 if (alt and (i // 3) % 2) else 1) for i in range(N)] for _ in range(6)]
            acc = [0] * N
            for r_, d_ in zip(rows, digs):
                fr = [center(x * NINV) for x in forward(r_)]
                fd = forward(d_)
                for k in range(N):
                    acc[k] = track(acc[k] + modmul(fd[k], fr[k]))
            out = inverse(acc)
            e1 = negacyclic(digs[0], rows[0])
            assert out == [6 * x for x in e1]
    print("ok: generator", g, "psi", PSI, "max |value| = 2^%.2f" % np.log2(stats["max_abs"]), "P/2 margin %.2f bits" % (np.log2(P / 2) - np.log2(max(abs(x) for x in exact))))
