"""CPU: the C-ABI library loads, exports every symbol include/rtfhe.h declares, fails loudly without a GPU,
and its host-side (non-GPU) entry points behave."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_all_exported():
    import rustfhe_amd as R
    lib = R.load()
    hdr = open(os.path.join(ROOT, "include", "rtfhe.h")).read()
    declared = set(re.findall(r"\b(rtfhe_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 25
    for sym in sorted(declared):
        assert hasattr(lib, sym), "include/rtfhe.h declares %s but librtfhe_hip.so does not export it" % sym
    assert declared == set(R._ffi.EXPORTED_SYMBOLS), declared ^ set(R._ffi.EXPORTED_SYMBOLS)


def test_header_is_plain_c99_and_a_c_host_links(tmp_path):
    """include/rtfhe.h is the drop-in boundary: plain C (no C++ or torch types), usable from a C host; without a GPU the
    library must refuse to create a context (no CPU fallback) -- checked here from C, linked against the built .so."""
    import shutil
    import subprocess
    if not shutil.which("gcc"):
        pytest.skip("gcc not available")
    src = tmp_path / "host.c"
    src.write_text('''#include <stdio.h>
#include "rtfhe.h"
int main(void) {
    rtfhe_params p; rtfhe_ctx *ctx = 0;
    rtfhe_default_params(&p);
    if (p.n != 635 || p.N != 1024 || p.l != 3 || p.bgbit != 6 || p.ks_t != 8 || p.ks_basebit != 2) return 2;
    int rc = rtfhe_ctx_create(&p, 0, &ctx);
    printf("%d %d %s\\n", rc, rtfhe_device_count(), rtfhe_version());
    if (rc == 0) { if (rtfhe_ctx_device_count(ctx) != 1) return 3; rtfhe_ctx_destroy(ctx); }
    /* the multi-device form: same refusal without a GPU, argument checks first */
    int ids[2] = {0, 0}; rtfhe_ctx *m = 0;
    if (rtfhe_ctx_create_multi(&p, ids, 65, &m) != RTFHE_ERR_INVALID || m) return 4;     /* more entries than a node has places */
    if (rtfhe_ctx_create_multi(&p, ids, 0, &m) != RTFHE_ERR_INVALID) return 5;
    int rcm = rtfhe_ctx_create_multi(&p, ids, 1, &m);
    if (rtfhe_device_count() == 0 ? (rcm != RTFHE_ERR_NO_DEVICE || m) : (rcm != 0 || rtfhe_ctx_device_count(m) != 1)) return 6;
    if (m) rtfhe_ctx_destroy(m);
    return 0;
}
''')
    import rustfhe_amd as R
    R.load()
    inc = os.path.join(ROOT, "include")
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", inc, "-fsyntax-only", str(src)])
    lib = R._ffi.lib_path()
    exe = tmp_path / "host"
    subprocess.check_call(["gcc", "-std=c99", "-I", inc, str(src), "-o", str(exe), lib,
                           "-Wl,-rpath," + os.path.dirname(lib), "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    rc, ndev = int(out.stdout.split()[0]), int(out.stdout.split()[1])
    assert (rc == 0) == (ndev > 0)                  # a context exists exactly when a HIP device does
    if ndev == 0:
        assert rc == R._ffi.ERR_NO_DEVICE


def test_library_is_a_gfx950_code_object():
    import rustfhe_amd as R
    R.load()
    blob = open(R._ffi.lib_path(), "rb").read()
    assert b"gfx950" in blob and b"k_bootstrap" in blob


def test_no_cpu_fallback_without_device():
    import rustfhe_amd as R
    if R.load().rtfhe_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(R.RtfheError) as ei:
        R.Engine(R.Params())
    assert ei.value.code == R._ffi.ERR_NO_DEVICE and "no CPU fallback" in str(ei.value)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "rustfhe_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".hpp", ".h")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "import orc" not in txt and "liboracle" not in txt and "tfhe_oracle" not in txt, f


def test_bad_params_rejected():
    import rustfhe_amd as R
    lib = R.load()
    h = C.c_void_p()
    for bad in (R.Params(N=512), R.Params(l=2, bgbit=10), R.Params(n=0), R.Params(n=5000), R.Params(N=1024, nbit=11)):
        assert lib.rtfhe_ctx_create(C.byref(bad), 0, C.byref(h)) in (R._ffi.ERR_INVALID, R._ffi.ERR_NO_DEVICE)
        assert not h.value
    assert lib.rtfhe_ctx_create(None, 0, C.byref(h)) == R._ffi.ERR_INVALID


def test_host_keygen_encrypt_decrypt_roundtrip():
    import rustfhe_amd as R
    p = R.Params(n=40)
    key0, key1, bk, ksk = R.keygen(p, 5, want_bk=False, want_ksk=True)
    assert set(np.unique(key0)) <= {0, 1} and set(np.unique(key1)) <= {0, 1} and len(key1) == p.N
    bits = np.array([0, 1, 1, 0, 1], np.uint8)
    ct = R.encrypt_bits(p, key0, bits, 9)
    assert ct.shape == (5, p.n + 1)
    assert np.array_equal(R.decrypt_bits(p, key0, ct), bits)
    ph = R.phases(p, key0, ct).astype(np.int64)
    exp = np.where(bits == 1, 0x20000000, 0xE0000000)
    assert np.abs(((ph - exp + 2 ** 31) % 2 ** 32) - 2 ** 31).max() < 2 ** 21       # sigma = 2^-15 -> 2^17 LSB
    # key-switch key rows decrypt to t * s_i * 2^(32 - 2(l+1))  (hom_nand/src/tlwe.rs:252-274)
    rows = ksk.reshape(p.N, p.ks_t, 3, p.n + 1)
    for (i, l, d) in [(0, 0, 0), (3, 2, 1), (17, 7, 2)]:
        got = int(R.phases(p, key0, rows[i, l, d][None])[0])
        want = (d + 1) * int(key1[i]) * (1 << (32 - 2 * (l + 1))) % 2 ** 32
        assert abs(((got - want + 2 ** 31) % 2 ** 32) - 2 ** 31) < 2 ** 21
    # same seed -> same keys
    k0b, _, _, _ = R.keygen(p, 5, want_bk=False, want_ksk=False)
    assert np.array_equal(key0, k0b)
    # the reference's container shape [[TLWERep; IKS_T = 4]; IKS_L = 8] (hom_nand/src/tlwe.rs:243-245): entries t = 1 .. 3 are the
    # compact key's rows, entry t = 4 encrypts 4 s_i / 4^(l+1) like KeySwitchingKey::new (tlwe.rs:252-274)
    for seed in (11, None):
        k4 = R.ksk_expand_ref(p, key0, key1, ksk, seed)
        assert k4.shape == (p.N, p.ks_t, 4, p.n + 1)
        assert np.array_equal(k4[:, :, :3], rows)
        for (i, l) in [(0, 0), (3, 2), (17, 7), (p.N - 1, 1)]:
            got = int(R.phases(p, key0, k4[i, l, 3][None])[0])
            want = 4 * int(key1[i]) * (1 << (32 - 2 * (l + 1))) % 2 ** 32
            assert abs(((got - want + 2 ** 31) % 2 ** 32) - 2 ** 31) < 2 ** 21
    assert np.array_equal(R.ksk_expand_ref(p, key0, key1, ksk, 11), R.ksk_expand_ref(p, key0, key1, ksk, 11))


def test_production_keygen_draws_from_the_os_csprng():
    """rtfhe_keygen / rtfhe_tlwe_encrypt_bits (no seed) are the production forms: OS CSPRNG + ChaCha20, never repeatable;
    the seeded xoshiro forms live under *_deterministic and are documented as test-only in include/rtfhe.h."""
    import rustfhe_amd as R
    p = R.Params(n=40)
    a = R.keygen(p, want_bk=False, want_ksk=True)
    b = R.keygen(p, want_bk=False, want_ksk=True)
    assert not np.array_equal(a[0], b[0]) or not np.array_equal(a[1], b[1])          # 1064 fresh key bits
    assert not np.array_equal(a[3], b[3])
    key0, key1, _, ksk = a
    assert set(np.unique(key0)) <= {0, 1} and 0.3 < key1.mean() < 0.7
    bits = np.array([1, 0, 1, 1, 0, 0, 1, 0], np.uint8)
    c1, c2 = R.encrypt_bits(p, key0, bits), R.encrypt_bits(p, key0, bits)
    assert not np.array_equal(c1, c2)                                                 # masks and noise are never reused
    assert np.array_equal(R.decrypt_bits(p, key0, c1), bits) and np.array_equal(R.decrypt_bits(p, key0, c2), bits)
    # masks look uniform over the f32-derived torus grid (utils/src/math.rs:425-432: 24 random bits, low byte zero)
    masks = c1[:, :p.n].reshape(-1)
    assert (masks & 0xff).max() == 0 and len(np.unique(masks)) == masks.size
    # key-switch rows are valid encryptions under the fresh keys (tlwe.rs:252-274)
    rows = ksk.reshape(p.N, p.ks_t, 3, p.n + 1)
    for (i, l, d) in [(1, 0, 0), (5, 3, 2)]:
        got = int(R.phases(p, key0, rows[i, l, d][None])[0])
        want = (d + 1) * int(key1[i]) * (1 << (32 - 2 * (l + 1))) % 2 ** 32
        assert abs(((got - want + 2 ** 31) % 2 ** 32) - 2 ** 31) < 2 ** 21
    hdr = open(os.path.join(ROOT, "include", "rtfhe.h")).read()
    assert "TEST ONLY" in hdr and "rtfhe_keygen_deterministic" in hdr


def test_host_bk_is_a_valid_trgsw_set(orc):
    """BK rows produced by the product's keygen decrypt (with the oracle as checker) to mu * 2^(32-6(i+1)) gadget rows."""
    import rustfhe_amd as R
    p = R.Params(n=3)
    key0, key1, bk, _ = R.keygen(p, 11, want_ksk=False)
    bk = bk.reshape(p.n, 2, 6, p.N)
    pl = orc.Plan(p.N)
    for i in range(p.n):
        for j in range(6):
            ct = np.concatenate([bk[i, 0, j], bk[i, 1, j]])
            ph = np.empty(p.N, np.uint32)
            orc.lib().orc_trlwe_phase(pl.h, p.N, key1.ctypes.data_as(C.POINTER(C.c_int32)), ct.ctypes.data_as(C.POINTER(C.c_uint32)),
                                      ph.ctypes.data_as(C.POINTER(C.c_uint32)))
            want = np.zeros(p.N, np.int64)
            g = int(key0[i]) << (32 - 6 * ((j % 3) + 1))
            if j < 3:
                want[0] = g                                # cipher[j] += mu / Bg^(j+1)
            else:
                # p_key[j] += g  <=>  phase = b - a*s gains  -g * s(X)
                want = (-g * key1.astype(np.int64)) % 2 ** 32
            err = ((ph.astype(np.int64) - want + 2 ** 31) % 2 ** 32) - 2 ** 31
            assert np.abs(err).max() < 2 ** 12, (i, j)


def test_shard_ranges_partition_a_batch():
    """The sharding arithmetic of a multi-device context (rtfhe_shard_range = what rtfhe_gate_batch / _mux_batch / _bootstrap_batch
    give device d): contiguous, in order, covering [0, count) exactly, sizes within one of each other, empty shards when
    count < n_dev; bad arguments are refused.  Pure host arithmetic: runs without a GPU."""
    import rustfhe_amd as R
    for n_dev in (1, 2, 3, 7, 8, 64):
        for count in (0, 1, 2, n_dev - 1, n_dev, n_dev + 1, 1000, 1024, 65536, 65537, 2 ** 33 + 5):
            if count < 0:
                continue
            edges = [R.shard_range(count, d, n_dev) for d in range(n_dev)]
            assert edges[0][0] == 0 and edges[-1][1] == count
            sizes = []
            for d, (b, e) in enumerate(edges):
                assert b <= e and (d == 0 or b == edges[d - 1][1])
                sizes.append(e - b)
            assert max(sizes) - min(sizes) <= 1 and sum(sizes) == count
    # the 8-GPU split of BASELINE config 3
    assert [R.shard_range(65536, d, 8) for d in range(8)] == [(8192 * d, 8192 * (d + 1)) for d in range(8)]
    for bad in ((10, -1, 4), (10, 4, 4), (10, 0, 0), (10, 0, 65)):
        with pytest.raises(R.RtfheError):
            R.shard_range(*bad)


def test_rust_sys_crate_declares_the_same_abi():
    """bindings/rust/rtfhe-sys is source only (no Rust toolchain here); keep its extern block in step with the header."""
    hdr = open(os.path.join(ROOT, "include", "rtfhe.h")).read()
    rs = open(os.path.join(ROOT, "bindings", "rust", "rtfhe-sys", "src", "lib.rs")).read()
    declared = set(re.findall(r"\b(rtfhe_[a-z0-9_]+)\s*\(", hdr))
    rust = set(re.findall(r"pub fn (rtfhe_[a-z0-9_]+)\s*\(", rs))
    assert declared == rust, declared ^ rust
    fields = re.findall(r"pub (\w+): i32", rs.split("pub struct rtfhe_params")[1].split("}")[0])
    assert fields == ["n", "N", "nbit", "l", "bgbit", "ks_t", "ks_basebit"]


def test_wire_format_roundtrip_and_corruption(tmp_path):
    import rustfhe_amd as R
    p = R.Params(n=12)
    key0, key1, bk, ksk = R.keygen(p, 77)
    path = str(tmp_path / "keys.rtfhe")
    R.save_keys(path, p, key0, key1, bk, ksk)
    assert os.path.getsize(path) == 8 + 28 + 8 + 4 * (p.n + p.N) + 4 * (p.bk_words + p.ksk_words) + 8
    q, k0, k1, b2, s2 = R.load_keys(path)
    assert (q.n, q.N, q.l, q.bgbit, q.ks_t, q.ks_basebit, q.nbit) == (12, 1024, 3, 6, 8, 2, 10)
    assert np.array_equal(k0, key0) and np.array_equal(k1, key1) and np.array_equal(b2, bk) and np.array_equal(s2, ksk)
    # public part only (what a server would hold): no secret keys in the file
    pub = str(tmp_path / "pub.rtfhe")
    R.save_keys(pub, p, None, None, bk, ksk)
    q, k0, k1, b2, s2 = R.load_keys(pub, want_ksk=False)
    assert k0 is None and k1 is None and s2 is None and np.array_equal(b2, bk)
    # ciphertext batches, including the empty one
    cts = R.encrypt_bits(p, key0, [1, 0, 1], 5)
    cpath = str(tmp_path / "batch.rtfhe")
    R.save_tlwe(cpath, p, cts)
    assert np.array_equal(R.load_tlwe(cpath), cts)
    R.save_tlwe(cpath, p, cts[:0])
    assert R.load_tlwe(cpath).shape == (0, p.n + 1)
    # corruption and truncation are detected
    raw = bytearray(open(path, "rb").read())
    raw[200] ^= 1
    open(path, "wb").write(raw)
    with pytest.raises(R.RtfheError):
        R.load_keys(path)
    open(path, "wb").write(raw[:1000])
    with pytest.raises(R.RtfheError):
        R.load_keys(path)
    with pytest.raises(R.RtfheError):
        R.load_tlwe(pub)


def test_shipped_twiddle_tables_are_the_golden_reference_tables():
    """rustfhe_amd/assets/twiddles_N*.bin (loaded through rtfhe_twiddles_load when a host's libm disagrees) hold exactly the tables
    of the reference build the golden vectors were made with, in the table-file format of rtfhe_wire.cpp, checksum included; and the
    oracle's own tables on THIS host agree with them (this image's libm is the reference build's)."""
    import struct
    import sys
    import numpy as np
    import rustfhe_amd as R
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc
    for N in (1024, 2048):
        path = R.engine.reference_twiddle_file(N)
        raw = open(path, "rb").read()
        assert raw[:8] == b"RTFHETW1" and struct.unpack("<ii", raw[8:16]) == (N, 0) and len(raw) == 16 + 4 * N * 8 + 8
        h = 0xcbf29ce484222325
        for x in raw[:-8]:
            h = ((h ^ x) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
        assert struct.unpack("<Q", raw[-8:])[0] == h
        g = np.load(os.path.join(ROOT, "tests", "golden", "fft_N%d.npz" % N))
        assert raw[16:16 + 2 * N * 8] == g["ifft_table"].tobytes() and raw[16 + 2 * N * 8:-8] == g["fft_table"].tobytes()
        a, b = orc.Plan(N).tables()
        assert a.tobytes() == g["ifft_table"].tobytes() and b.tobytes() == g["fft_table"].tobytes()
    L = R.load()
    assert L.rtfhe_twiddles_load(None, b"/nonexistent", None) == R._ffi.ERR_INVALID
    assert L.rtfhe_twiddles_write(None, b"/nonexistent") == R._ffi.ERR_INVALID


def test_twiddle_table_file_roundtrip_and_corruption(tmp_path):
    """rtfhe_twiddles_file_write / _read (host-only file I/O of the table format): the shipped file reads back as the golden tables, a written
    file is byte-identical to the shipped one, and a wrong degree, a flipped bit or a truncated file is refused."""
    import numpy as np
    import rustfhe_amd as R
    L = R.load()
    N = 1024
    g = np.load(os.path.join(ROOT, "tests", "golden", "fft_N%d.npz" % N))
    a, b = np.empty(2 * N), np.empty(2 * N)
    shipped = R.engine.reference_twiddle_file(N)
    assert L.rtfhe_twiddles_file_read(os.fsencode(shipped), N, a.ctypes.data, b.ctypes.data) == 0
    assert a.tobytes() == g["ifft_table"].tobytes() and b.tobytes() == g["fft_table"].tobytes()
    path = str(tmp_path / "tw.bin")
    assert L.rtfhe_twiddles_file_write(os.fsencode(path), N, a.ctypes.data, b.ctypes.data) == 0
    assert open(path, "rb").read() == open(shipped, "rb").read()
    assert L.rtfhe_twiddles_file_read(os.fsencode(path), 2048, np.empty(4096).ctypes.data, np.empty(4096).ctypes.data) == R._ffi.ERR_INVALID
    raw = bytearray(open(path, "rb").read())
    raw[1000] ^= 1
    open(path, "wb").write(bytes(raw))
    assert L.rtfhe_twiddles_file_read(os.fsencode(path), N, a.ctypes.data, b.ctypes.data) == R._ffi.ERR_INVALID
    open(path, "wb").write(bytes(raw[:500]))
    assert L.rtfhe_twiddles_file_read(os.fsencode(path), N, a.ctypes.data, b.ctypes.data) == R._ffi.ERR_INVALID
