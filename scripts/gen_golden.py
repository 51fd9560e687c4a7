#!/usr/bin/env python3
"""Generates tests/golden/*.npz in THIS container from the reference's own compiled native FFT
(oracle/_ref/libspqlios_ref.so, built by `make -C oracle ref` from /root/reference sources in place)
driven by the oracle's restated Rust glue (oracle/tfhe_oracle.c with the reference FFT plugged in
through orc_plan_set_hooks).

Fixtures are data only (inputs + expected outputs); keys are regenerated from the recorded seed by
the oracle's deterministic keygen, their fnv64 hashes are stored to detect drift.

    python scripts/gen_golden.py            # N = 1024 fixtures (+ spawns itself for N = 2048, 16)
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def ref_tables(R, N):
    """Raw bytes of the reference's twiddle tables (new_ifft_table / new_fft_table structs:
    {uint64 n; double *trig; double *data; void *buf}, spqlios-fft-impl.cpp:48-60)."""
    out = []
    for fn in (R.new_ifft_table, R.new_fft_table):
        st = fn(N)
        trig = C.cast(st + 8, C.POINTER(C.c_void_p))[0]
        arr = np.ctypeslib.as_array(C.cast(trig, C.POINTER(C.c_double)), shape=(2 * N,)).copy()
        arr[2 * N - 8:] = 0.0   # the reference never writes the last 8 entries
        out.append(arr)
    return out


def fft_vectors(N, seed):
    R = orc.ref_lib()
    h = R.Spqlios_new(N)
    rng = np.random.default_rng(seed)
    srcs = np.stack([
        rng.integers(-32, 32, N), rng.integers(-2 ** 31, 2 ** 31, N), rng.integers(0, 2, N),
        np.eye(1, N, 1)[0] + np.eye(1, N, 2)[0],
    ]).astype(np.int32)
    fwd = np.empty((len(srcs), N), np.float64)
    for i, s in enumerate(srcs):
        R.Spqlios_ifft_i32(h, fwd[i].ctypes.data, s.ctypes.data)
    spec = np.stack([fwd[0] * 12345.0, fwd[1] * 3.0, fwd[2] * 1048576.0, fwd[3]])
    inv = np.empty((len(spec), N), np.uint32)
    for i, s in enumerate(spec):
        s = np.ascontiguousarray(s)
        R.Spqlios_fft_u32(h, inv[i].ctypes.data, s.ctypes.data)
    ifft_t, fft_t = ref_tables(R, N)
    return dict(fft_src=srcs, fft_fwd=fwd, inv_src=spec, inv_out=inv, ifft_table=ifft_t, fft_table=fft_t)


def small(N):
    np.savez_compressed(os.path.join(GOLD, "fft_N%d.npz" % N), **fft_vectors(N, 1000 + N))
    print("wrote fft_N%d.npz" % N)


def main():
    os.makedirs(GOLD, exist_ok=True)
    assert orc.have_ref(), "build oracle/_ref first (make -C oracle ref)"
    if len(sys.argv) > 1:
        small(int(sys.argv[1]))
        return
    for N in (16, 2048):   # one N per process: the reference caches 2/N in a function-local static
        subprocess.check_call([sys.executable, __file__, str(N)])
    small(1024)

    P = orc.Params()
    seed = 20211003
    keys = orc.Keys(P, seed)
    plan_ref = orc.Plan(P.N).use_reference_fft()
    rng = orc.Rng()
    orc.lib().orc_rng_seed(C.byref(rng), 4242)
    bits0 = [0, 0, 1, 1, 1, 0, 1, 1, 0, 1]
    bits1 = [0, 1, 0, 1, 1, 1, 0, 0, 0, 1]
    ops = [orc.NAND] * 4 + [orc.AND, orc.OR, orc.XOR, orc.NOT, orc.XOR, orc.OR]
    in0 = keys.encrypt_bits(bits0, rng=rng)
    in1 = keys.encrypt_bits(bits1, rng=rng)
    outs = np.stack([orc.gate(P, plan_ref, op, keys.bk_f, None, keys.ksk, a, b) for op, a, b in zip(ops, in0, in1)])
    dec = keys.decrypt_bits(outs)
    truth = {orc.NAND: lambda a, b: 1 - (a & b), orc.AND: lambda a, b: a & b, orc.OR: lambda a, b: a | b,
             orc.XOR: lambda a, b: a ^ b, orc.NOT: lambda a, b: 1 - a}
    assert dec == [truth[o](a, b) for o, a, b in zip(ops, bits0, bits1)], dec
    # stage vectors on gate 0's pre-combined input
    t0 = orc.gate_linear(P, orc.NAND, in0[0], in1[0])
    acc3 = orc.blind_rotate(P, plan_ref, keys.bk_f, None, t0, steps=3)
    acc_full = orc.blind_rotate(P, plan_ref, keys.bk_f, None, t0)
    ext = orc.sample_extract(P, acc_full)
    ks = orc.key_switch(P, keys.ksk, ext)
    assert np.array_equal(ks, outs[0])
    trlwe = np.random.default_rng(5).integers(0, 2 ** 32, (2, 2 * P.N), dtype=np.uint64).astype(np.uint32)
    ep_idx = np.array([0, 417], np.int32)
    ep = np.stack([orc.external_product(P, plan_ref, keys.bk_f[i * P.trgsw_words:(i + 1) * P.trgsw_words], None, t)
                   for i, t in zip(ep_idx, trlwe)])
    mux_out = orc.mux(P, plan_ref, keys.bk_f, None, keys.ksk, in0[2], in0[0], in1[1])
    np.savez_compressed(
        os.path.join(GOLD, "gate_N1024.npz"),
        seed=np.uint64(seed), ops=np.array(ops, np.int32), bits0=np.array(bits0, np.uint8), bits1=np.array(bits1, np.uint8),
        in0=in0, in1=in1, out=outs, key0=keys.key0, key1=keys.key1,
        bk_t_fnv=np.uint64(orc.fnv64(keys.bk_t)), bk_f_fnv=np.uint64(orc.fnv64(keys.bk_f)), ksk_fnv=np.uint64(orc.fnv64(keys.ksk)),
        acc_steps3=acc3, acc_full=acc_full, extract=ext, ep_idx=ep_idx, ep_in=trlwe, ep_out=ep, mux_out=mux_out)
    print("wrote gate_N1024.npz; decrypted:", dec)


if __name__ == "__main__":
    main()
