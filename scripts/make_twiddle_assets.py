#!/usr/bin/env python3
"""Writes rustfhe_amd/assets/twiddles_N{1024,2048}.bin -- the twiddle tables of the reference build the golden vectors were made with
(tests/golden/fft_N*.npz: 'ifft_table' / 'fft_table', dumped by scripts/gen_golden.py from the reference's compiled new_ifft_table /
new_fft_table in the build container) -- in the table-file format of rtfhe_twiddles_load (rustfhe_amd/csrc/rtfhe_wire.cpp):
    "RTFHETW1" | i32 N | i32 0 | f64 ifft_table[2N] | f64 fft_table[2N] | u64 fnv1a of all previous bytes
The tables are data (libm outputs), not code: SURVEY H5 asks for exactly this -- "dump them from the reference build and commit as fixtures"."""
import os
import struct
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def fnv1a(b):
    h = 0xcbf29ce484222325
    for x in b:
        h = ((h ^ x) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
    return h


for N in (1024, 2048):
    g = np.load(os.path.join(ROOT, "tests", "golden", "fft_N%d.npz" % N))
    body = b"RTFHETW1" + struct.pack("<ii", N, 0) + g["ifft_table"].astype("<f8").tobytes() + g["fft_table"].astype("<f8").tobytes()
    assert len(body) == 16 + 4 * N * 8
    path = os.path.join(ROOT, "rustfhe_amd", "assets", "twiddles_N%d.bin" % N)
    with open(path, "wb") as f:
        f.write(body + struct.pack("<Q", fnv1a(body)))
    print(path, len(body) + 8, "bytes")
