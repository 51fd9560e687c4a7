"""ctypes binding of librtfhe_hip.so (the C ABI declared in include/rtfhe.h).

There is no fallback: if the shared library is missing it is built with hipcc; if that is impossible
an ImportError/RuntimeError is raised.  Nothing here ever touches oracle/.
"""
import ctypes as C
import os

from . import build as _build

HERE = os.path.dirname(os.path.abspath(__file__))


class Params(C.Structure):
    """rtfhe_params; defaults = the reference's constants (tlwe.rs:175-180, trlwe.rs:76, trgsw.rs:112-115, tfhe.rs:16)."""
    _fields_ = [(k, C.c_int32) for k in ("n", "N", "nbit", "l", "bgbit", "ks_t", "ks_basebit")]

    def __init__(self, n=635, N=1024, nbit=None, l=3, bgbit=6, ks_t=8, ks_basebit=2):
        super().__init__()
        self.n, self.N, self.l, self.bgbit, self.ks_t, self.ks_basebit = n, N, l, bgbit, ks_t, ks_basebit
        self.nbit = nbit if nbit is not None else int(N).bit_length() - 1

    @property
    def bk_words(self):
        return self.n * 2 * 2 * self.l * self.N

    @property
    def ksk_words(self):
        return self.N * self.ks_t * ((1 << self.ks_basebit) - 1) * (self.n + 1)


class PeerInfo(C.Structure):
    """rtfhe_peer_info: what the runtime reported about entry d of a multi-device context against the primary (include/rtfhe.h)."""
    _fields_ = [("device", C.c_int32), ("same_device", C.c_int32), ("can_access_from_primary", C.c_int32), ("can_access_to_primary", C.c_int32),
                ("enabled_from_primary", C.c_int32), ("enabled_to_primary", C.c_int32), ("link_type", C.c_uint32), ("hops", C.c_uint32),
                ("scatter_ms", C.c_float), ("compute_ms", C.c_float), ("gather_ms", C.c_float)]

    LINK_NAMES = {1: "HyperTransport", 2: "QPI", 3: "PCIe", 4: "InfiniBand", 5: "xGMI", 0xFFFFFFFF: "not reported"}

    def as_dict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_}
        d["link"] = self.LINK_NAMES.get(self.link_type, "type %d" % self.link_type)
        for k in ("scatter_ms", "compute_ms", "gather_ms"):
            d[k] = None if d[k] < 0 else round(d[k], 4)
        return d


NAND, AND, OR, XOR, NOT, COPY, ANDNY = range(7)
BACKEND_FFT64_MIRROR, BACKEND_NTT_EXACT, BACKEND_FFT_SPLIT_EXACT = 0, 1, 2
OK, ERR_INVALID, ERR_NO_DEVICE, ERR_HIP, ERR_STATE, ERR_NOMEM = 0, -1, -2, -3, -4, -5

_SIGNATURES = {
    "rtfhe_default_params": (None, ["PP"]),
    "rtfhe_ctx_create": (C.c_int, ["PP", C.c_int, C.POINTER(C.c_void_p)]),
    "rtfhe_shard_range": (C.c_int, [C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "rtfhe_ctx_create_multi": (C.c_int, ["PP", C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)]),
    "rtfhe_ctx_device_count": (C.c_int, [C.c_void_p]),
    "rtfhe_ctx_memory_bytes": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_size_t)]),
    "rtfhe_ctx_peer_info": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(PeerInfo)]),
    "rtfhe_device_link": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "rtfhe_ctx_destroy": (None, [C.c_void_p]),
    "rtfhe_host_alloc": (C.c_void_p, [C.c_size_t]),
    "rtfhe_host_free": (None, [C.c_void_p]),
    "rtfhe_last_error": (C.c_char_p, [C.c_void_p]),
    "rtfhe_version": (C.c_char_p, []),
    "rtfhe_device_count": (C.c_int, []),
    "rtfhe_set_backend": (C.c_int, [C.c_void_p, C.c_int]),
    "rtfhe_get_backend": (C.c_int, [C.c_void_p]),
    "rtfhe_get_twiddles": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "rtfhe_set_twiddles": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "rtfhe_ctx_params": (C.c_int, [C.c_void_p, "PP"]),
    "rtfhe_twiddles_load": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_int32)]),
    "rtfhe_twiddles_write": (C.c_int, [C.c_void_p, C.c_char_p]),
    "rtfhe_twiddles_file_write": (C.c_int, [C.c_char_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "rtfhe_twiddles_file_read": (C.c_int, [C.c_char_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "rtfhe_load_bk_torus": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rtfhe_load_bk_fft": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rtfhe_export_bk_fft": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rtfhe_load_ksk": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rtfhe_load_ksk_ref": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rtfhe_gate_batch": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "rtfhe_mux_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "rtfhe_bootstrap_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "rtfhe_gate_batch_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "rtfhe_mux_batch_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "rtfhe_bootstrap_batch_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "rtfhe_circuit_wave_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p]),
    "rtfhe_circuit_create": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int32), C.c_int32,
                                       C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "rtfhe_circuit_launch": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rtfhe_circuit_destroy": (None, [C.c_void_p]),
    "rtfhe_sync": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rtfhe_timer_begin": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rtfhe_timer_end": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "rtfhe_timer_end_detail": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "rtfhe_blind_rotate_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_size_t]),
    "rtfhe_external_product_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "rtfhe_key_switch_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "rtfhe_ifft_i32_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "rtfhe_fft_u32_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "rtfhe_ifft_f64_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "rtfhe_fft_f64_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "rtfhe_poly_mul_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "rtfhe_fft_plan_create": (C.c_int, [C.c_int32, C.c_int, C.POINTER(C.c_void_p)]),
    "rtfhe_fft_plan_destroy": (None, [C.c_void_p]),
    "rtfhe_fft_plan_degree": (C.c_int32, [C.c_void_p]),
    "rtfhe_fft_plan_get_twiddles": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "rtfhe_fft_plan_set_twiddles": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "rtfhe_fft_plan_ifft_i32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "rtfhe_fft_plan_ifft_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "rtfhe_fft_plan_fft_u32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "rtfhe_fft_plan_fft_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "rtfhe_fft_plan_poly_mul": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "rtfhe_keygen": (C.c_int, ["PP", C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rtfhe_keygen_with_keys": (C.c_int, ["PP", C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rtfhe_tlwe_encrypt_bits": (C.c_int, ["PP", C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "rtfhe_ksk_expand_ref": (C.c_int, ["PP", C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rtfhe_ksk_expand_ref_deterministic": (C.c_int, ["PP", C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rtfhe_keygen_deterministic": (C.c_int, ["PP", C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rtfhe_keygen_with_keys_deterministic": (C.c_int, ["PP", C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rtfhe_tlwe_encrypt_bits_deterministic": (C.c_int, ["PP", C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_size_t]),
    "rtfhe_tlwe_decrypt_bits": (C.c_int, ["PP", C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "rtfhe_tlwe_phase": (C.c_int, ["PP", C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "rtfhe_keys_write": (C.c_int, [C.c_char_p, "PP", C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rtfhe_keys_read_header": (C.c_int, [C.c_char_p, "PP", C.POINTER(C.c_uint32)]),
    "rtfhe_keys_read": (C.c_int, [C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rtfhe_tlwe_write": (C.c_int, [C.c_char_p, C.c_int32, C.c_void_p, C.c_size_t]),
    "rtfhe_tlwe_read": (C.c_int, [C.c_char_p, C.POINTER(C.c_int32), C.POINTER(C.c_uint64), C.c_void_p, C.c_size_t]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib = None


def lib_path():
    return _build.LIB


def load(build_if_missing=True):
    """Loads (building first if needed) librtfhe_hip.so.  Raises if it cannot be had."""
    global _lib
    if _lib is not None:
        return _lib
    try:
        # PyTorch wheels bundle their own HIP runtime; two HIP runtimes in one process do not coexist (whichever loads
        # second sees no device).  Loading torch's first lets librtfhe_hip.so bind to it, so both can be used together.
        import torch  # noqa: F401
    except Exception:
        pass
    override = os.environ.get("RTFHE_LIB")          # a prebuilt library to load as it is (A/B runs: no staleness check, no rebuild)
    if override:
        path = os.path.abspath(override)
        if not os.path.exists(path):
            raise ImportError("RTFHE_LIB=%s does not exist" % override)
        L = C.CDLL(path)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = [C.POINTER(Params) if a == "PP" else a for a in args]
        _lib = L
        return L
    if build_if_missing:
        _build.build()
    if not os.path.exists(_build.LIB):
        raise ImportError("librtfhe_hip.so is missing and could not be built (hipcc required); "
                          "rustfhe_amd has no CPU fallback")
    L = C.CDLL(_build.LIB)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(L, name)   # AttributeError here = ABI mismatch: fail loudly
        fn.restype = res
        fn.argtypes = [C.POINTER(Params) if a == "PP" else a for a in args]
    _lib = L
    return L
