"""CPU: the model the split-FFT exact backend was designed on (scripts/xfft/model.py) -- run as a test so that the claims the device code rests on
stay checked: exactness of the split products on random and worst-case inputs at N = 1024 and 2048, the proven error bound below 1/2, the
N = 2048 decomposition into two 512-point halves per transform (what k_bootstrap_xquad's waves compute) against the whole transform, and the
instruction counts bench.py prices the kernels with."""
import importlib.util
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _model():
    spec = importlib.util.spec_from_file_location("xfft_model", os.path.join(ROOT, "scripts", "xfft", "model.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_model_runs_clean():
    assert _model().main() == 0


def test_error_bound_has_margin_at_both_degrees():
    m = _model()
    assert m.error_bound(1024) < 2.0 ** -8 and m.error_bound(2048) < 2.0 ** -6
    # the bound scales with the magnitude of the key halves: an UNSPLIT 32-bit key could not be proven exact (this is why the key is split)
    assert m.error_bound(1024, half_max=2.0 ** 31) > 0.5


def test_split_key_halves_recombine_and_stay_in_range():
    m = _model()
    k = np.array([0, 1, 0x7FFF, 0x8000, 0xFFFF, 0x10000, 0x7FFF8000, 0x7FFFFFFF, 0x80000000, 0x80008000, 0xFFFFFFFF], np.uint32)
    hi, lo = m.split_key(k)
    assert np.all((hi << 16) + lo == k.view(np.int32).astype(np.int64))
    assert lo.min() >= -2 ** 15 and lo.max() < 2 ** 15 and np.abs(hi).max() <= 2 ** 15      # hi reaches +2^15 exactly once (0x7FFF8000)


def test_device_link_and_peer_info_validate_their_arguments_without_a_gpu():
    """rtfhe_device_link / rtfhe_ctx_peer_info (round 6) answer errors, never crash, when there is nothing to ask (no GPU in the CPU suite)."""
    import ctypes as C
    import rustfhe_amd as R
    L = R.load()
    can, lt, hops = C.c_int32(), C.c_uint32(), C.c_uint32()
    assert L.rtfhe_device_link(0, 1, None, C.byref(lt), C.byref(hops)) == R._ffi.ERR_INVALID
    rc = L.rtfhe_device_link(0, 1, C.byref(can), C.byref(lt), C.byref(hops))
    assert rc in (R._ffi.ERR_NO_DEVICE, R._ffi.ERR_INVALID)          # no device here; on a one-GPU box: ids out of range
    info = R._ffi.PeerInfo()
    assert L.rtfhe_ctx_peer_info(None, 1, C.byref(info)) == R._ffi.ERR_INVALID
