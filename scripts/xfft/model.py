#!/usr/bin/env python3
"""Model of the split-FFT exact backend (RTFHE_BACKEND_FFT_SPLIT_EXACT), run on the CPU before anything goes to HIP.

What it computes.  The external product needs, per output polynomial, S = sum_{r<6} D_r (*) K_r mod 2^32 (negacyclic, D_r = digit polynomials in
[-32, 31], K_r = key rows as signed 32-bit words; utils/src/math.rs:238-257 is the exact product the oracle's `exact_int` backend restates).
An FP64 FFT cannot return S exactly (|S| < 2^48.6 against a transform error of ~0.3: the reference's +-1 LSB).  Split every key
word once, at key-load time, into signed 16-bit halves  K = 2^16 hi + lo,  lo in [-2^15, 2^15), hi in [-2^15, 2^15]:

    S = 2^16 * (sum_r D_r (*) hi_r)  +  (sum_r D_r (*) lo_r)          each sum |.| <= 6 * N * 32 * 2^15 = 2^33.6 (N = 1024)

Each sum comes out of an FMA-contracted FP64 FFT with an error far below 1/2 (bound below), is rounded to the nearest integer by a magic-number
addition, and the two are recombined mod 2^32 in integer registers.  Bit-identical to exact integer arithmetic for EVERY input, not on average.

The transform (n = N/2 complex points; ring C[X]/(X^n - i), folded input z_j = a_j + i a_{j+n}):
  forward = Cooley-Tukey butterflies (a + w b, a - w b), natural order in -> bit-reversed out, the twist MERGED into the twiddles
            (stage s, block B: w = exp(i theta/2), theta_root = pi/2, children theta/2 and theta/2 + pi) -- no twist pass at all
  inverse = the same butterfly form as a standard radix-2 DIT on the bit-reversed spectrum (first three stages: twiddles 1, -i, (+-1-i)/sqrt 2),
            then ONE untwist multiply by psi^-j / n fused with the rounding
Both directions use the 6-FMA butterfly  a' = fma(wr, br, fma(-wi, bi, ar)) ...,  b' = fma(2, a, -a').

Run: python3 scripts/xfft/model.py            (asserts exactness on random and worst-case inputs, the error bound, prints instruction counts)
"""
import sys
import numpy as np

U = 2.0 ** -53


def brv(x, bits):
    return int(format(x, "0%db" % bits)[::-1], 2) if bits else 0


class Plan:
    def __init__(self, N, root_theta=np.pi / 2):
        self.N, self.n = N, N // 2
        self.L = self.n.bit_length() - 1
        n, L = self.n, self.L
        # forward: theta[s][B], s = 1..L
        self.fw = []
        th = np.array([root_theta], dtype=np.longdouble)
        for s in range(1, L + 1):
            half = th / 2
            self.fw.append(np.exp(1j * half.astype(np.float64)).astype(np.complex128) if False else
                           (np.cos(half) + 1j * np.sin(half)).astype(np.complex128))
            nxt = np.empty(2 * len(th), dtype=np.longdouble)
            nxt[0::2] = half
            nxt[1::2] = half + np.longdouble(np.pi)
            th = nxt
        self.leaf_theta = th                      # root of position p: exp(i leaf_theta[p]) = psi^(4 brv(p) + 1)
        # inverse DIT: stage t = 1..L, twiddle omega_{2^t}^{-q}, q < 2^(t-1)
        self.iw = []
        for t in range(1, L + 1):
            q = np.arange(1 << (t - 1), dtype=np.longdouble)
            a = -2 * np.longdouble(np.pi) * q / (1 << t)
            self.iw.append((np.cos(a) + 1j * np.sin(a)).astype(np.complex128))
        j = np.arange(n, dtype=np.longdouble)
        a = -(root_theta / n) * j                 # psi^-j with psi = exp(i root_theta / n)
        self.untw = ((np.cos(a) + 1j * np.sin(a)) / n).astype(np.complex128)

    def forward(self, z):
        z = np.array(z, dtype=np.complex128)
        n = self.n
        for s in range(1, self.L + 1):
            m = n >> (s - 1)
            v = z.reshape(-1, 2, m // 2)
            w = self.fw[s - 1][:, None]
            t = w * v[:, 1, :]
            a = v[:, 0, :] + t
            b = 2 * v[:, 0, :] - a                # the device's b' = fma(2, a, -a')
            z = np.stack([a, b], axis=1).reshape(-1)
        return z

    def inverse(self, z):
        """-> n complex values y_j (already untwisted and scaled): coefficient j = Re, coefficient j + n = Im"""
        z = np.array(z, dtype=np.complex128)
        n = self.n
        for t in range(1, self.L + 1):
            half = 1 << (t - 1)
            v = z.reshape(-1, 2, half)
            w = self.iw[t - 1][None, :]
            tt = w * v[:, 1, :]
            a = v[:, 0, :] + tt
            b = 2 * v[:, 0, :] - a
            z = np.stack([a, b], axis=1).reshape(-1)
        return z * self.untw


def fold(a, n):
    a = np.asarray(a, dtype=np.float64)
    return a[:n] + 1j * a[n:]


def negacyclic_exact(d, k):
    """exact integer negacyclic product of int64 vectors (|result| < 2^62 for the sizes here)"""
    N = len(d)
    full = np.convolve(d.astype(np.int64), k.astype(np.int64))
    out = full[:N].copy()
    out[: N - 1] -= full[N:]
    return out


def split_key(k_u32):
    k = k_u32.astype(np.uint32).view(np.int32).astype(np.int64)
    lo = ((k + 0x8000) & 0xFFFF) - 0x8000
    hi = (k - lo) >> 16
    assert np.all(lo >= -0x8000) and np.all(lo < 0x8000) and np.all(np.abs(hi) <= 0x8000) and np.all((hi << 16) + lo == k)
    return hi, lo


def cross_split_fft(plan, digits, key_rows):
    """digits: [rows][N] ints in [-32, 31]; key_rows: [rows][N] uint32 -> (u32 result, max distance from an integer before rounding)"""
    n = plan.n
    acc_hi = np.zeros(n, dtype=np.complex128)
    acc_lo = np.zeros(n, dtype=np.complex128)
    for d, k in zip(digits, key_rows):
        hi, lo = split_key(k)
        fd = plan.forward(fold(d, n))
        acc_hi += fd * plan.forward(fold(hi, n))
        acc_lo += fd * plan.forward(fold(lo, n))
    worst = 0.0
    parts = []
    for acc in (acc_hi, acc_lo):
        y = plan.inverse(acc)
        v = np.concatenate([y.real, y.imag])
        r = np.rint(v)
        worst = max(worst, float(np.max(np.abs(v - r))))
        parts.append(r.astype(np.int64))
    res = ((parts[0] << 16) + parts[1]) & 0xFFFFFFFF
    return res.astype(np.uint32), worst


def cross_exact(digits, key_rows):
    N = len(digits[0])
    s = np.zeros(N, dtype=object)
    for d, k in zip(digits, key_rows):
        ks = k.astype(np.uint32).view(np.int32).astype(np.int64)
        hi, lo = split_key(k)
        s = s + (negacyclic_exact(np.asarray(d), hi).astype(object) << 16) + negacyclic_exact(np.asarray(d), lo).astype(object)
        del ks
    return np.array([int(x) & 0xFFFFFFFF for x in s], dtype=np.uint32)


# ---- N = 2048 on the device: a 1024-point transform as TWO 512-point transforms, one per wave (rtfhe_kernels_xfft2.hpp) ------------------------
# forward: stage 1 pairs z_j with z_{j+512} under ONE block twiddle c = exp(i pi/4); the "top" results (+) continue in the ring X^512 - c, the
# "bottom" ones (-) in X^512 + c: two independent 512-point transforms of the N = 1024 shape with root angles pi/4 and pi/4 + pi -- wave h = 0 / 1
# -- whose outputs are positions [512 h, 512 h + 512) of the full spectrum.  inverse: nine DIT stages inside each half, then the last stage across
# the halves fused with the untwist:  y_q = T_q U_q + B_q V_q,  y_{q+512} = (T_q U_q - B_q V_q) e^{-i pi/4},  U_q = psi^-q / n,
# V_q = omega^-q U_q  (psi = exp(i (pi/2) / n), omega = exp(2 pi i / n), n = 1024).
def forward_halves(z):
    """-> the full 1024-point spectrum computed as the device does (two halves)"""
    c = np.exp(1j * np.pi / 4)
    out = []
    for h in (0, 1):
        sub = Plan(1024, root_theta=np.pi / 4 + h * np.pi)
        t = z[:512] + (c if h == 0 else -c) * z[512:]
        out.append(sub.forward(t))
    return np.concatenate(out)


def inverse_halves(spec):
    """-> 1024 complex values y_j (coefficient j = Re, coefficient j + 1024 = Im), untwisted and scaled, computed as the device does"""
    n = 1024
    sub = Plan(1024)                                   # only its inverse stage twiddles are used (the standard radix-2 DIT ones)
    parts = []
    for h in (0, 1):
        z = np.array(spec[512 * h:512 * h + 512], dtype=np.complex128)
        for t in range(1, 10):
            half = 1 << (t - 1)
            v = z.reshape(-1, 2, half)
            tt = sub.iw[t - 1][None, :] * v[:, 1, :]
            a = v[:, 0, :] + tt
            z = np.stack([a, 2 * v[:, 0, :] - a], axis=1).reshape(-1)
        parts.append(z)
    T, B = parts
    q = np.arange(512, dtype=np.longdouble)
    ua = -(np.longdouble(np.pi) / 2 / n) * q
    U = ((np.cos(ua) + 1j * np.sin(ua)) / n).astype(np.complex128)
    va = ua - 2 * np.longdouble(np.pi) * q / n
    V = ((np.cos(va) + 1j * np.sin(va)) / n).astype(np.complex128)
    y_lo = T * U + B * V
    D = T * U - B * V
    s = np.sqrt(0.5)
    y_hi = s * (D.real + D.imag) + 1j * s * (D.imag - D.real)
    return np.concatenate([y_lo, y_hi])


def check_halves(rng):
    plan = Plan(2048)
    z = rng.integers(-32, 32, 1024) + 1j * rng.integers(-32, 32, 1024)
    full = plan.forward(z)
    assert np.max(np.abs(forward_halves(z) - full)) < 1e-9 * np.max(np.abs(full)), "forward halves"
    spec = full * plan.forward(rng.integers(-2 ** 15, 2 ** 15, 1024) + 1j * rng.integers(-2 ** 15, 2 ** 15, 1024))
    ref = plan.inverse(spec)
    got = inverse_halves(spec)
    assert np.max(np.abs(got - ref)) < 1e-9 * np.max(np.abs(ref)), "inverse halves"


def error_bound(N, rows=6, digit_max=32, half_max=2.0 ** 15):
    """Worst-case |computed - exact| of one rounded sum, every input (see the derivation in the module docstring of rtfhe_xfft.hpp):
    relative l2 error per butterfly stage <= 5u (two nested FMAs per component, b' = 2a - a', twiddle rounding), L stages per transform;
    the digit spectrum, the key spectrum (transformed by the same device code) and the inverse each carry L * 5u, the multiply-accumulate 4u,
    the untwist 3u; l-infinity <= l2; ||D||_1 <= N * digit_max, ||K||_2 <= half_max * sqrt N."""
    L = (N // 2).bit_length() - 1
    stage = 5 * U
    d1, k2 = N * digit_max, half_max * np.sqrt(N)
    spectrum_side = rows * (2 * L * stage + 4 * U) * d1 * k2
    inverse_side = rows * (L * stage + 3 * U) * d1 * k2
    grid = 2.0 ** -15          # the two roundings to the 2^-16 grid of the fused untwist (magic constant 1.5 * 2^36 + 0.5)
    return spectrum_side + inverse_side + grid


def instruction_counts(N):
    """FP64-rate wave instructions per CMUX (two waves at N = 1024: each owns one polynomial's three digit rows) beside the mirror's 3,744 and the
    NTT backend's 6,648.  N = 2048: four waves (polynomial x half of the spectrum), beside the mirror's 8,112 and the NTT backend's 19,968."""
    if N == 2048:
        fwd = 3 * 12 * 6
        inv9 = (4 * 4 + 4 * 4 + 2 * 4 + 2 * 6) + 2 * 12 * 6
        per_wave = {
            "int -> f64 of the digits and their stage-1 sums / differences (4 per point)": 3 * 8 * 4,
            "stage 1 across the halves (one block twiddle: 2 FMA per point)": 3 * 8 * 2,
            "forward transforms of the half (3)": 3 * fwd,
            "multiply-accumulate (12 row-halves x 8 points x 4 FMA)": 12 * 8 * 4,
            "nine inverse stages inside the half (hi, lo)": 2 * inv9,
            "last inverse stage across the halves fused with untwist and rounding (20 per pair of outputs, 4 pairs)": 2 * 4 * 20,
        }
        return per_wave, 4 * sum(per_wave.values())
    assert N == 1024
    R = 8
    butterflies_per_pass = 12                   # three radix-2 stages on 8 points
    fwd = 3 * butterflies_per_pass * 6          # 216, no twist
    inv = (4 * 4 + 4 * 4 + 2 * 4 + 2 * 6) + 2 * butterflies_per_pass * 6      # pass 1 with twiddles 1, -i, (+-1-i)/sqrt2 = 52; passes 2, 3 = 144
    per_wave = {
        "int -> f64 of the digits": 3 * 2 * R,
        "forward transforms (3)": 3 * fwd,
        "multiply-accumulate (12 row-halves x 8 points x 4 FMA)": 12 * R * 4,
        "inverse transforms (hi, lo)": 2 * inv,
        "untwist fused with rounding (2 FMA per real output)": 2 * 2 * R * 2,
    }
    return per_wave, 2 * sum(per_wave.values())


def main():
    rng = np.random.default_rng(2026)
    ok = True
    for N in (1024, 2048):
        plan = Plan(N)
        bound = error_bound(N)
        assert bound < 2.0 ** -6, bound
        # 1. a single product against the definition
        d = rng.integers(-32, 32, N)
        k = rng.integers(0, 2 ** 32, N, dtype=np.uint64).astype(np.uint32)
        got, w = cross_split_fft(plan, [d], [k])
        assert np.array_equal(got, cross_exact([d], [k])), "single product"
        # 2. six rows, random
        worst = 0.0
        for trial in range(4):
            D = [rng.integers(-32, 32, N) for _ in range(6)]
            K = [rng.integers(0, 2 ** 32, N, dtype=np.uint64).astype(np.uint32) for _ in range(6)]
            got, w = cross_split_fft(plan, D, K)
            worst = max(worst, w)
            assert np.array_equal(got, cross_exact(D, K)), "random rows"
        # 3. worst-case magnitudes: all digits -32 / +31 / alternating, key words 0x80000000, 0x7FFFFFFF, 0x8000FFFF, 0x7FFF8000, sign patterns that
        # align every term of one output coefficient
        worst_wc = 0.0
        pats = [np.full(N, -32), np.full(N, 31), np.where(np.arange(N) % 2 == 0, -32, 31), np.where(np.arange(N) < N // 2, -32, 31)]
        keys = [np.full(N, 0x80000000, np.uint32), np.full(N, 0x7FFFFFFF, np.uint32), np.full(N, 0x8000FFFF, np.uint32),
                np.full(N, 0x7FFF8000, np.uint32), np.where(np.arange(N) < N // 2, 0x80008000, 0x7FFF7FFF).astype(np.uint32),
                np.where(np.arange(N) % 2 == 0, 0x80008000, 0x7FFF7FFF).astype(np.uint32)]
        for dp in pats:
            for kp in keys:
                D = [dp] * 6
                K = [kp] * 6
                got, w = cross_split_fft(plan, D, K)
                worst_wc = max(worst_wc, w)
                assert np.array_equal(got, cross_exact(D, K)), "worst-case rows"
        # the aligned case: output coefficient N-1 of (-32, ..., -32) (*) (-2^15, ...) has all N terms of one sign
        assert worst < bound and worst_wc < bound, (worst, worst_wc, bound)
        print("N = %4d: exact on random and worst-case inputs; distance from an integer before rounding: random %.3g (2^%.1f), worst-case patterns "
              "%.3g (2^%.1f); proven bound %.3g (2^%.1f) < 1/2" % (N, worst, np.log2(worst), worst_wc, np.log2(max(worst_wc, 1e-300)), bound, np.log2(bound)))
    check_halves(rng)
    print("N = 2048 as two 512-point halves per transform (forward: one block twiddle across the halves; inverse: last stage fused with the untwist): agrees with the whole transform")
    per_wave, per_cmux = instruction_counts(1024)
    for k, v in per_wave.items():
        print("  %-60s %5d per wave and step" % (k, v))
    print("FP64-rate instructions per CMUX (N = 1024): %d   (fft64 mirror 3,744; NTT backend 6,648)" % per_cmux)
    assert per_cmux < 5000
    per_wave, per_cmux = instruction_counts(2048)
    for k, v in per_wave.items():
        print("  %-104s %5d per wave and step" % (k, v))
    print("FP64-rate instructions per CMUX (N = 2048): %d   (fft64 mirror 8,112; NTT backend 19,968)" % per_cmux)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
