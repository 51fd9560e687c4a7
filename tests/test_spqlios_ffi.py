"""The reference's only existing FFI -- Spqlios_new / _destructor / _ifft / _ifft_u32 / _ifft_i32 / _fft / _fft_u32 / _poly_mul
(utils/src/spqlios.rs:18-32, spqlios-wrapper.cpp:9-53) -- exported BY NAME from librtfhe_hip.so (include/rtfhe_spqlios.h), so the
reference's `utils` crate can link against the engine unchanged."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = ["Spqlios_new", "Spqlios_destructor", "Spqlios_ifft", "Spqlios_ifft_u32", "Spqlios_ifft_i32", "Spqlios_fft",
         "Spqlios_fft_u32", "Spqlios_poly_mul"]


def test_header_declares_the_reference_ffi_and_the_library_exports_it():
    import rustfhe_amd as R
    hdr = open(os.path.join(ROOT, "include", "rtfhe_spqlios.h")).read()
    declared = re.findall(r"\b(Spqlios_[a-z0-9_]+)\s*\(", hdr.split("typedef struct SpqliosImpl")[1])
    assert declared == NAMES
    ref_rs = os.path.join("/root/reference", "utils", "src", "spqlios.rs")
    if os.path.exists(ref_rs):          # this container only: the names the reference's crate binds
        assert re.findall(r"fn (Spqlios_[a-z0-9_]+)\(", open(ref_rs).read()) == NAMES
    L = R.load()
    for n in NAMES:
        getattr(L, n)
    # plain C99 header
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-fsyntax-only", "-x", "c",
                           os.path.join(ROOT, "include", "rtfhe_spqlios.h")])


def test_unsupported_degree_or_no_device_gives_null():
    import rustfhe_amd as R
    L = R.load()
    L.Spqlios_new.restype = C.c_void_p
    L.Spqlios_new.argtypes = [C.c_int32]
    L.Spqlios_destructor.argtypes = [C.c_void_p]
    for bad in (0, 8, 48, 1000, 4096):              # not a power of two in [16, 2048]: the reference aborts in require(), this returns NULL
        assert L.Spqlios_new(bad) is None
    if L.rtfhe_device_count() <= 0:
        assert L.Spqlios_new(16) is None and L.Spqlios_new(1024) is None          # no GPU: no handle, and no CPU fallback
        h = C.c_void_p()
        assert L.rtfhe_fft_plan_create(16, 0, C.byref(h)) == R._ffi.ERR_NO_DEVICE and not h.value
    L.Spqlios_destructor(None)                      # harmless
    L.rtfhe_fft_plan_destroy(None)


@pytest.mark.gpu
@pytest.mark.parametrize("N", [16, 32, 512, 1024, 2048])      # 16: the size of the reference's own unit test (spqlios.rs:243-276)
def test_c_host_gets_the_reference_bytes_through_both_libraries(tmp_path, N):
    ref = os.path.join(ROOT, "oracle", "_ref", "libspqlios_ref.so")
    if not os.path.exists(ref):
        pytest.skip("oracle/_ref/libspqlios_ref.so not built (needs the reference checkout at build time)")
    import rustfhe_amd.build as b
    exe = str(tmp_path / "spqlios_ffi")
    subprocess.check_call(["gcc", "-O1", "-std=gnu99", "-Wall", os.path.join(ROOT, "tests", "c", "spqlios_ffi_main.c"), "-o", exe, "-ldl"])
    out = subprocess.run([exe, b.LIB, ref, str(N)], capture_output=True, text=True, timeout=300)     # one N per process (H7)
    assert out.returncode == 0 and "spqlios ffi ok" in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("N", [16, 64, 256, 512])
def test_transform_plan_at_every_size_matches_the_oracle(N, orc):
    """rtfhe_fft_plan (the engine's transforms off the gate path's two sizes) against the oracle's restatement of ifft_model / fft_model,
    itself pinned to the reference's compiled spqlios at N = 16 / 1024 / 2048 (tests/test_oracle_vs_ref.py): every byte."""
    import numpy as np
    import rustfhe_amd as R
    rng = np.random.default_rng(N)
    plan, pl = R.FftPlan(N), orc.Plan(N)
    try:
        assert np.array_equal(np.concatenate(plan.get_twiddles()), np.concatenate(pl.tables()))
        digits = rng.integers(-32, 32, (5, N)).astype(np.int32)
        words = rng.integers(0, 2 ** 32, (5, N), dtype=np.uint64).astype(np.uint32)
        f = plan.ifft_i32(np.concatenate([digits, words.view(np.int32)]))
        exp = np.stack([pl.ifft_i32(r) for r in np.concatenate([digits, words.view(np.int32)])])
        assert f.tobytes() == exp.tobytes()
        spectra = exp * rng.integers(1, 1000, exp.shape)
        assert np.array_equal(plan.fft_u32(spectra), np.stack([pl.fft_u32(r) for r in spectra]))
        assert plan.fft_f64(spectra).tobytes() == np.stack([pl.fft_f64(r) for r in spectra]).tobytes()
        halves = digits.astype(np.float64) * 0.5
        assert plan.ifft_f64(halves).tobytes() == np.stack([pl.ifft_f64(r) for r in halves]).tobytes()
        # round trip fft_torus(ifft_torus(p)): exact for small polynomials (the reference's "step 1", spqlios.rs:243-261, uses bits 1); for
        # full 32-bit words the truncation toward zero turns a rounding error of -epsilon into -1 -- the same words as the oracle's either way
        back = plan.fft_u32(plan.ifft_i32(words.view(np.int32)))
        assert np.array_equal(back, np.stack([pl.fft_u32(pl.ifft_i32(r)) for r in words.view(np.int32)]))
        assert np.abs((back - words).view(np.int32)).max() <= 1
        small = rng.integers(0, 64, (5, N)).astype(np.uint32)
        got = plan.poly_mul(words, small)
        want = np.stack([orc.negacyclic_mul(a, b.astype(np.int32)) for a, b in zip(words, small)])
        assert np.abs((got - want).view(np.int32)).max() <= 1       # the FFT product is the exact one +- 1 LSB
    finally:
        plan.close()


@pytest.mark.gpu
def test_reference_fft_test_kat_and_golden_vectors_at_n16_on_the_gpu():
    """The reference's own unit test of this FFI (utils/src/spqlios.rs:243-276): fft_torus(ifft_torus(X + X^2)) is exact and
    poly_mul(X + X^2, X + X^2) = X^2 + 2 X^3 + X^4 -- through the Spqlios_* symbols of the engine at N = 16; then the golden transform
    vectors made with the reference's compiled spqlios (tests/golden/fft_N16.npz) through the HIP path."""
    import numpy as np
    import rustfhe_amd as R
    L = R.load()
    L.Spqlios_new.restype = C.c_void_p
    L.Spqlios_new.argtypes = [C.c_int32]
    for f in ("Spqlios_destructor", "Spqlios_ifft_u32", "Spqlios_fft_u32", "Spqlios_poly_mul", "Spqlios_ifft_i32"):
        getattr(L, f).restype = None
    L.Spqlios_destructor.argtypes = [C.c_void_p]
    L.Spqlios_ifft_u32.argtypes = L.Spqlios_fft_u32.argtypes = L.Spqlios_ifft_i32.argtypes = [C.c_void_p] * 3
    L.Spqlios_poly_mul.argtypes = [C.c_void_p] * 4
    h = L.Spqlios_new(16)
    assert h
    try:
        pol = np.zeros(16, np.uint32)
        pol[1] = pol[2] = 1
        spec, back, prod = np.empty(16, np.float64), np.empty(16, np.uint32), np.empty(16, np.uint32)
        L.Spqlios_ifft_u32(h, spec.ctypes.data, pol.ctypes.data)
        L.Spqlios_fft_u32(h, back.ctypes.data, spec.ctypes.data)
        assert np.array_equal(back, pol), "fft_test: step 1"
        L.Spqlios_poly_mul(h, prod.ctypes.data, pol.ctypes.data, pol.ctypes.data)
        expect = np.zeros(16, np.uint32)
        expect[2], expect[3], expect[4] = 1, 2, 1
        assert np.abs((prod - expect).view(np.int32)).max() < 1000, "fft_test: step 2 (very_close)"
    finally:
        L.Spqlios_destructor(h)
    g = np.load(os.path.join(ROOT, "tests", "golden", "fft_N16.npz"))
    plan = R.FftPlan(16)
    try:
        ifft_t, fft_t = plan.get_twiddles()
        assert ifft_t.tobytes() == g["ifft_table"].tobytes() and fft_t.tobytes() == g["fft_table"].tobytes()
        assert plan.ifft_i32(g["fft_src"]).tobytes() == g["fft_fwd"].tobytes()
        assert np.array_equal(plan.fft_u32(g["inv_src"]), g["inv_out"])
    finally:
        plan.close()
