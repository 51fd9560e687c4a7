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
    assert L.Spqlios_new(16) is None and L.Spqlios_new(1000) is None
    if L.rtfhe_device_count() <= 0:
        assert L.Spqlios_new(1024) is None          # no GPU: no handle, and no CPU fallback
    L.Spqlios_destructor.argtypes = [C.c_void_p]
    L.Spqlios_destructor(None)                      # harmless


@pytest.mark.gpu
@pytest.mark.parametrize("N", [1024, 2048])
def test_c_host_gets_the_reference_bytes_through_both_libraries(tmp_path, N):
    ref = os.path.join(ROOT, "oracle", "_ref", "libspqlios_ref.so")
    if not os.path.exists(ref):
        pytest.skip("oracle/_ref/libspqlios_ref.so not built (needs the reference checkout at build time)")
    import rustfhe_amd.build as b
    exe = str(tmp_path / "spqlios_ffi")
    subprocess.check_call(["gcc", "-O1", "-std=gnu99", "-Wall", os.path.join(ROOT, "tests", "c", "spqlios_ffi_main.c"), "-o", exe, "-ldl"])
    out = subprocess.run([exe, b.LIB, ref, str(N)], capture_output=True, text=True, timeout=300)     # one N per process (H7)
    assert out.returncode == 0 and "spqlios ffi ok" in out.stdout, out.stdout + out.stderr
