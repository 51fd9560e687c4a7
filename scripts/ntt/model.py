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

# Renormalisation points (stage counts), round 5: ONE inside the forward transform (after stage 7 = the third stage of pass 2; the outputs go to
# the multiply-accumulate unnormalised) and three inside the inverse (after stages 4, 8 and 10) plus one of the input sums.  Round 4 had (5, 10) and
# (3, 6, 9, 10): 384 FP64-rate instructions per CMUX more.  worst_case_bounds() below proves the schedule for EVERY input, not just the sampled ones.
FWD_NORM = (7,)
INV_NORM = (4, 8, 10)
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

def worst_case_bounds(rows=6):
    """Interval propagation of |value| through forward transform -> multiply-accumulate over `rows` key rows -> inverse transform, for ANY input:
    doubles hold the values exactly as long as every |value| < 2^53.  modmul(a, w) with |w| <= P/2 returns r = a w - q P with
    |q - a w / P| <= 1/2 + 3 * 2^-53 |a w| / P (three roundings: fl(a w), the constant 1/P, their product), i.e. |r| <= P/2 + 3 |a| |w| / 2^53 + 1;
    normalize(x) leaves |x'| <= P/2 + 3 |x| / 2^53 + 1.  Returns the largest bound met."""
    LIM = 1 << 53
    half = P // 2 + 1
    def mm(a):                     # bound of modmul(a, w), |w| <= P/2
        return half + (3 * a * half >> 53) + 1
    def nz(a):
        return half + (3 * a >> 53) + 1
    worst = 0
    def chk(b):
        nonlocal worst
        assert b < LIM, "bound %.3f P reaches 2^53" % (b / P)
        worst = max(worst, b)
        return b
    # stages 1 and 2 act on decomposition digits through tables of centred residues: x = a + z1 c +- (z2 b + z2 z1 d), |a| <= 32, every product <= P/2
    b = chk(32 + 3 * half)
    for stage in range(3, 11):     # Cooley-Tukey: t = modmul(x1, z); x0 +- t
        b = chk(b + mm(chk(b)))
        if stage in FWD_NORM:
            b = nz(b)
    fwd_out = b
    acc = chk(rows * mm(fwd_out))  # the key rows are stored normalised; products are summed unreduced
    b = nz(acc)
    for stage in range(1, 11):     # Gentleman-Sande: x0 = u + v; x1 = modmul(u - v, z)
        b = max(chk(2 * b), mm(chk(2 * b)))
        if stage in INV_NORM:
            b = nz(b)
    assert b <= half + 2
    return {"forward_out_over_P": fwd_out / P, "mac_sum_over_P": acc / P, "worst_over_P": worst / P, "limit_over_P": LIM / P}


def normalize(x):
    q = int(np.rint(np.float64(float(x)) * np.float64(PINV)))
    return track(x - q * P)

def forward_digits(a):
    """the device's forward transform of a DIGIT polynomial: stages 1 and 2 from the five digit tables (rtfhe_ntt.hpp, first_two_stages_digits),
    stages 3..10 as butterflies"""
    assert all(-32 <= v < 32 for v in a)
    z1, z2, z3 = ZETA[1], ZETA[2], ZETA[3]
    tab = [[center(d * c) for d in range(-32, 32)] for c in (z1, z2, z3, z2 * z1 % P, z3 * z1 % P)]
    x = [0] * N
    for j in range(N // 4):
        ai, b, c, d = a[j], a[j + 256], a[j + 512], a[j + 768]
        zc, z2b, z3b, z2d, z3d = tab[0][c + 32], tab[1][b + 32], tab[2][b + 32], tab[3][d + 32], tab[4][d + 32]
        s_, t_, p_, q_ = track(ai + zc), track(ai - zc), track(z2b + z2d), track(z3b - z3d)
        x[j], x[j + 256], x[j + 512], x[j + 768] = track(s_ + p_), track(s_ - p_), track(t_ + q_), track(t_ - q_)
    return forward(x, first_stage=3)


def forward(a, first_stage=1):
    a = list(a)
    h = N // 2 >> (first_stage - 1)
    stage = first_stage - 1
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
    assert all((x - y) % P == 0 for x, y in zip(fa, forward_digits(a))), "table stages != butterfly stages"
    for p in (0, 1, 5, 1023):
        root = pow(PSI, 2 * brv(p, LOGN) + 1, P)
        assert (fa[p] - sum(c * pow(root, i, P) for i, c in enumerate(a))) % P == 0
    # external-product-like accumulation: 6 rows, torus x digits, N^-1 folded into the key rows
    rows = [[random.randrange(-2 ** 31, 2 ** 31) for _ in range(N)] for _ in range(6)]
    digs = [[random.randrange(-32, 32) for _ in range(N)] for _ in range(6)]
    acc = [0] * N
    for r_, d_ in zip(rows, digs):
        fr = [center(x * NINV) for x in forward(r_)]
        fd = forward_digits(d_)
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
            digs = [[min(31, dval * (-1 if (alt and (i // 3) % 2) else 1)) for i in range(N)] for _ in range(6)]      # digits live in [-32, 31]
            acc = [0] * N
            for r_, d_ in zip(rows, digs):
                fr = [center(x * NINV) for x in forward(r_)]
                fd = forward_digits(d_)
                for k in range(N):
                    acc[k] = track(acc[k] + modmul(fd[k], fr[k]))
            out = inverse(acc)
            e1 = negacyclic(digs[0], rows[0])
            assert out == [6 * x for x in e1]
    wb = worst_case_bounds()
    print("worst-case bounds (any input): forward output <= %.3f P, sum of 6 products <= %.3f P, largest intermediate %.3f P < 2^53 = %.6f P"
          % (wb["forward_out_over_P"], wb["mac_sum_over_P"], wb["worst_over_P"], wb["limit_over_P"]))
    print("ok: generator", g, "psi", PSI, "max |value| = 2^%.2f" % np.log2(stats["max_abs"]), "P/2 margin %.2f bits" % (np.log2(P / 2) - np.log2(max(abs(x) for x in exact))))
