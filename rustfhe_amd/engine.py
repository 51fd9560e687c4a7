"""Engine: one rtfhe_ctx (one GPU) behind numpy / torch-device-pointer calls.

Every compute method runs the HIP kernels of librtfhe_hip.so through the C ABI; errors surface as
RtfheError with the library's message.  No CPU path exists in this package.
"""
import ctypes as C
import os

import numpy as np

from . import _ffi
from ._ffi import AND, ANDNY, COPY, ERR_INVALID, NAND, NOT, OR, XOR, Params  # noqa: F401


class RtfheError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("rtfhe error %d: %s" % (code, msg))
        self.code = code


def _np(a, dtype):
    a = np.ascontiguousarray(a, dtype=dtype)
    return a


def _ptr(a):
    return C.c_void_p(a.ctypes.data) if a is not None else None


ASSETS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets")


def reference_twiddle_file(N):
    """The shipped twiddle tables of the reference build the golden vectors were made with (rustfhe_amd/assets/), or None."""
    path = os.path.join(ASSETS, "twiddles_N%d.bin" % N)
    return path if os.path.exists(path) else None


class Engine:
    def __init__(self, params=None, device=0, devices=None, reference_twiddles=True):
        """device: one GPU (rtfhe_ctx_create).  devices=[d0, d1, ...]: one context over several GPUs of the node
        (rtfhe_ctx_create_multi): keys are loaded once and replicated device-to-device, the host-buffer batch calls shard
        contiguous gate ranges over them, the *_dev batch calls shard a batch resident on devices[0]; netlists and stage-level calls stay on devices[0].
        reference_twiddles: the context builds its twiddle tables with this host's libm; when that libm disagrees with the shipped
        tables of the reference build (a few entries may be an ulp apart between libms) the shipped tables are installed
        instead (rtfhe_twiddles_load) and self.twiddle_entries_replaced says how many entries differed.  False: this host's libm as it is."""
        self.L = _ffi.load()
        self.p = params or Params()
        h = C.c_void_p()
        if devices is not None:
            ids = (C.c_int * len(devices))(*devices)
            rc = self.L.rtfhe_ctx_create_multi(C.byref(self.p), ids, len(devices), C.byref(h))
            device = devices[0] if len(devices) else 0
        else:
            rc = self.L.rtfhe_ctx_create(C.byref(self.p), device, C.byref(h))
        if rc != 0:
            raise RtfheError(rc, (self.L.rtfhe_last_error(None) or b"").decode())
        self.h = h
        self.device = device
        self.twiddle_entries_replaced = 0
        path = reference_twiddle_file(self.p.N) if reference_twiddles else None
        if path:
            self.twiddle_entries_replaced = self.twiddles_load(path)

    def twiddles_load(self, path):
        """rtfhe_twiddles_load: installs the file's tables if they differ from the context's; returns the number of differing entries."""
        n = C.c_int32(0)
        self._ck(self.L.rtfhe_twiddles_load(self.h, os.fsencode(path), C.byref(n)))
        return n.value

    def twiddles_write(self, path):
        self._ck(self.L.rtfhe_twiddles_write(self.h, os.fsencode(path)))

    def device_count(self):
        return self.L.rtfhe_ctx_device_count(self.h)

    def close(self):
        if getattr(self, "h", None):
            self.L.rtfhe_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != 0:
            raise RtfheError(rc, (self.L.rtfhe_last_error(self.h) or b"").decode())

    # ---- keys -------------------------------------------------------------------------------
    def load_bk_torus(self, bk):
        bk = _np(bk, np.uint32).reshape(-1)
        assert bk.size == self.p.bk_words, "bk must be u32[n][2][2l][N]"
        self._ck(self.L.rtfhe_load_bk_torus(self.h, _ptr(bk)))

    def load_bk_fft(self, bk_f):
        bk_f = _np(bk_f, np.float64).reshape(-1)
        assert bk_f.size == self.p.bk_words, "bk_f must be f64[n][2][2l][N]"
        self._ck(self.L.rtfhe_load_bk_fft(self.h, _ptr(bk_f)))

    def export_bk_fft(self):
        out = np.empty(self.p.bk_words, np.float64)
        self._ck(self.L.rtfhe_export_bk_fft(self.h, _ptr(out)))
        return out

    def load_ksk(self, ksk):
        ksk = _np(ksk, np.uint32).reshape(-1)
        assert ksk.size == self.p.ksk_words, "ksk must be u32[N][t][base-1][n+1]"
        self._ck(self.L.rtfhe_load_ksk(self.h, _ptr(ksk)))

    def load_ksk_ref(self, ksk_ref):
        """The reference's own shape, KeySwitchingKey(Vec<[[TLWERep; 4]; 8]>) flattened to u32[N][t][base][n+1]
        (hom_nand/src/tlwe.rs:243-245); the never-read entry t = base of every level is dropped by the library."""
        ksk_ref = _np(ksk_ref, np.uint32).reshape(-1)
        base = 1 << self.p.ks_basebit
        assert ksk_ref.size == self.p.ksk_words // (base - 1) * base, "ksk_ref must be u32[N][t][base][n+1]"
        self._ck(self.L.rtfhe_load_ksk_ref(self.h, _ptr(ksk_ref)))

    def set_backend(self, backend):
        self._ck(self.L.rtfhe_set_backend(self.h, backend))

    def backend(self):
        return self.L.rtfhe_get_backend(self.h)

    def twiddles(self):
        a = np.zeros(2 * self.p.N, np.float64)
        b = np.zeros(2 * self.p.N, np.float64)
        self._ck(self.L.rtfhe_get_twiddles(self.h, _ptr(a), _ptr(b)))
        return a, b

    def set_twiddles(self, ifft_table, fft_table):
        a, b = _np(ifft_table, np.float64), _np(fft_table, np.float64)
        assert a.size == 2 * self.p.N and b.size == 2 * self.p.N
        self._ck(self.L.rtfhe_set_twiddles(self.h, _ptr(a), _ptr(b)))

    # ---- hot path, host buffers -------------------------------------------------------------
    def gate_batch(self, op, in0, in1=None):
        in0 = _np(in0, np.uint32).reshape(-1, self.p.n + 1)
        if in1 is not None:
            in1 = _np(in1, np.uint32).reshape(-1, self.p.n + 1)
            assert in1.shape == in0.shape
        out = np.empty_like(in0)
        self._ck(self.L.rtfhe_gate_batch(self.h, op, _ptr(in0), _ptr(in1), _ptr(out), in0.shape[0]))
        return out

    def mux_batch(self, c, in0, in1):
        c = _np(c, np.uint32).reshape(-1, self.p.n + 1)
        in0 = _np(in0, np.uint32).reshape(c.shape)
        in1 = _np(in1, np.uint32).reshape(c.shape)
        out = np.empty_like(c)
        self._ck(self.L.rtfhe_mux_batch(self.h, _ptr(c), _ptr(in0), _ptr(in1), _ptr(out), c.shape[0]))
        return out

    def bootstrap_batch(self, tlwe):
        tlwe = _np(tlwe, np.uint32).reshape(-1, self.p.n + 1)
        out = np.empty_like(tlwe)
        self._ck(self.L.rtfhe_bootstrap_batch(self.h, _ptr(tlwe), _ptr(out), tlwe.shape[0]))
        return out

    # ---- hot path, device buffers (torch tensors or raw pointers) ---------------------------
    @staticmethod
    def _dev(t):
        if t is None:
            return None
        if isinstance(t, int):
            return C.c_void_p(t)
        return C.c_void_p(t.data_ptr())

    def gate_batch_dev(self, op, d_in0, d_in1, d_out, count, stream=None):
        self._ck(self.L.rtfhe_gate_batch_dev(self.h, op, self._dev(d_in0), self._dev(d_in1), self._dev(d_out),
                                             count, C.c_void_p(stream) if stream else None))

    def mux_batch_dev(self, d_c, d_in0, d_in1, d_out, count, stream=None):
        self._ck(self.L.rtfhe_mux_batch_dev(self.h, self._dev(d_c), self._dev(d_in0), self._dev(d_in1), self._dev(d_out),
                                            count, C.c_void_p(stream) if stream else None))

    def bootstrap_batch_dev(self, d_tlwe, d_out, count, stream=None):
        self._ck(self.L.rtfhe_bootstrap_batch_dev(self.h, self._dev(d_tlwe), self._dev(d_out), count, C.c_void_p(stream) if stream else None))

    def memory_bytes(self, d=0):
        """Device memory entry d of the context holds right now: keys in every form built so far, staging and scratch (not the twiddle tables, not
        live circuits' sample buffers)."""
        b = C.c_size_t()
        self._ck(self.L.rtfhe_ctx_memory_bytes(self.h, d, C.byref(b)))
        return b.value

    def peer_info(self, d):
        """What the runtime reported about entry d >= 1 of a multi-device context against the primary (peer access both ways, whether enabling it
        worked, link type and hops) and the phases of that entry's share of the last device-resident sharded batch; a dict (rtfhe_peer_info)."""
        info = _ffi.PeerInfo()
        self._ck(self.L.rtfhe_ctx_peer_info(self.h, d, C.byref(info)))
        return info.as_dict()

    def circuit_wave_dev(self, d_ops, d_idx0, d_idx1, d_idx_out, d_wires, num_wires, count, stream=None):
        """One dependency wave of a netlist; wire indices / opcodes are validated on the device against num_wires
        (rows of d_wires) and a violation surfaces as RtfheError at the next sync()."""
        self._ck(self.L.rtfhe_circuit_wave_dev(self.h, self._dev(d_ops), self._dev(d_idx0), self._dev(d_idx1),
                                               self._dev(d_idx_out), self._dev(d_wires), num_wires, count,
                                               C.c_void_p(stream) if stream else None))

    def circuit_create(self, d_ops, d_idx0, d_idx1, d_idx_out, wave_offsets, d_wires, num_wires):
        """Records the waves [wave_offsets[w], wave_offsets[w+1]) of a levelised netlist into one HIP graph; returns a handle
        for circuit_launch / circuit_destroy.  The device arrays must stay alive while the handle exists."""
        offs = _np(wave_offsets, np.int32)
        h = C.c_void_p()
        self._ck(self.L.rtfhe_circuit_create(self.h, self._dev(d_ops), self._dev(d_idx0), self._dev(d_idx1), self._dev(d_idx_out),
                                             offs.ctypes.data_as(C.POINTER(C.c_int32)), offs.size - 1, self._dev(d_wires), num_wires,
                                             C.byref(h)))
        return h

    def circuit_launch(self, circuit, stream=None):
        self._ck(self.L.rtfhe_circuit_launch(circuit, C.c_void_p(stream) if stream else None))

    def circuit_destroy(self, circuit):
        self.L.rtfhe_circuit_destroy(circuit)

    def sync(self, stream=None):
        self._ck(self.L.rtfhe_sync(self.h, C.c_void_p(stream) if stream else None))

    def timer_begin(self, stream=None):
        self._ck(self.L.rtfhe_timer_begin(self.h, C.c_void_p(stream) if stream else None))

    def timer_end(self, stream=None):
        ms, n = C.c_double(), C.c_int64()
        self._ck(self.L.rtfhe_timer_end(self.h, C.c_void_p(stream) if stream else None, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def timer_end_detail(self, stream=None):
        """(total ms, ms of it inside the batch key switches of the split path, kernel launches)"""
        ms, ks, n = C.c_double(), C.c_double(), C.c_int64()
        self._ck(self.L.rtfhe_timer_end_detail(self.h, C.c_void_p(stream) if stream else None, C.byref(ms), C.byref(ks), C.byref(n)))
        return ms.value, ks.value, n.value

    # ---- stage level ------------------------------------------------------------------------
    def blind_rotate_batch(self, tlwe, steps=None):
        tlwe = _np(tlwe, np.uint32).reshape(-1, self.p.n + 1)
        acc = np.empty((tlwe.shape[0], 2, self.p.N), np.uint32)
        self._ck(self.L.rtfhe_blind_rotate_batch(self.h, _ptr(tlwe), self.p.n if steps is None else steps,
                                                 _ptr(acc), tlwe.shape[0]))
        return acc

    def external_product_batch(self, bk_index, trlwe):
        trlwe = _np(trlwe, np.uint32).reshape(-1, 2, self.p.N)
        idx = _np(bk_index, np.int32).reshape(-1)
        assert idx.size == trlwe.shape[0]
        out = np.empty_like(trlwe)
        self._ck(self.L.rtfhe_external_product_batch(self.h, _ptr(idx), _ptr(trlwe), _ptr(out), trlwe.shape[0]))
        return out

    def key_switch_batch(self, tlwe1):
        tlwe1 = _np(tlwe1, np.uint32).reshape(-1, self.p.N + 1)
        out = np.empty((tlwe1.shape[0], self.p.n + 1), np.uint32)
        self._ck(self.L.rtfhe_key_switch_batch(self.h, _ptr(tlwe1), _ptr(out), tlwe1.shape[0]))
        return out

    def ifft_i32_batch(self, src):
        src = _np(src, np.int32).reshape(-1, self.p.N)
        res = np.empty(src.shape, np.float64)
        self._ck(self.L.rtfhe_ifft_i32_batch(self.h, _ptr(src), _ptr(res), src.shape[0]))
        return res

    def ifft_f64_batch(self, src):
        """Spqlios_ifft: forward transform of double polynomials."""
        src = _np(src, np.float64).reshape(-1, self.p.N)
        res = np.empty(src.shape, np.float64)
        self._ck(self.L.rtfhe_ifft_f64_batch(self.h, _ptr(src), _ptr(res), src.shape[0]))
        return res

    def fft_f64_batch(self, src):
        """Spqlios_fft: inverse transform to doubles (scaled by 2/N, no truncation)."""
        src = _np(src, np.float64).reshape(-1, self.p.N)
        res = np.empty(src.shape, np.float64)
        self._ck(self.L.rtfhe_fft_f64_batch(self.h, _ptr(src), _ptr(res), src.shape[0]))
        return res

    def poly_mul_batch(self, a, b):
        """Spqlios_poly_mul: negacyclic product of torus polynomials through the FP64 transform."""
        a = _np(a, np.uint32).reshape(-1, self.p.N)
        b = _np(b, np.uint32).reshape(a.shape)
        res = np.empty_like(a)
        self._ck(self.L.rtfhe_poly_mul_batch(self.h, _ptr(a), _ptr(b), _ptr(res), a.shape[0]))
        return res

    def fft_u32_batch(self, src):
        src = _np(src, np.float64).reshape(-1, self.p.N)
        res = np.empty(src.shape, np.uint32)
        self._ck(self.L.rtfhe_fft_u32_batch(self.h, _ptr(src), _ptr(res), src.shape[0]))
        return res


class FftPlan:
    """rtfhe_fft_plan: the reference's two transforms (and Spqlios_poly_mul) at any power of two 16 <= N <= 2048 on the GPU
    (FFT_Processor_Spqlios(N), utils/src/spqlios/fft_processor_spqlios.cpp:7-14); numpy in / numpy out.  The gate path's N = 1024 /
    2048 have their own, fast kernels behind Engine; this serves every other size the reference's FFT FFI accepts."""

    def __init__(self, N, device=0):
        self.L = _ffi.load()
        self.N = int(N)
        h = C.c_void_p()
        rc = self.L.rtfhe_fft_plan_create(self.N, device, C.byref(h))
        if rc != 0:
            raise RtfheError(rc, (self.L.rtfhe_last_error(None) or b"").decode())
        self.h = h

    def _ck(self, rc):
        if rc != 0:
            raise RtfheError(rc, (self.L.rtfhe_last_error(None) or b"").decode())

    def close(self):
        if getattr(self, "h", None):
            self.L.rtfhe_fft_plan_destroy(self.h)
            self.h = None

    __del__ = close

    def _run(self, fn, src, in_dtype, out_dtype, src2=None):
        src = _np(src, in_dtype).reshape(-1, self.N)
        res = np.empty(src.shape, out_dtype)
        if src2 is None:
            self._ck(fn(self.h, _ptr(src), _ptr(res), src.shape[0]))
        else:
            src2 = _np(src2, in_dtype).reshape(src.shape)
            self._ck(fn(self.h, _ptr(src), _ptr(src2), _ptr(res), src.shape[0]))
        return res

    def ifft_i32(self, src):
        return self._run(self.L.rtfhe_fft_plan_ifft_i32, src, np.int32, np.float64)

    def ifft_f64(self, src):
        return self._run(self.L.rtfhe_fft_plan_ifft_f64, src, np.float64, np.float64)

    def fft_u32(self, src):
        return self._run(self.L.rtfhe_fft_plan_fft_u32, src, np.float64, np.uint32)

    def fft_f64(self, src):
        return self._run(self.L.rtfhe_fft_plan_fft_f64, src, np.float64, np.float64)

    def poly_mul(self, a, b):
        return self._run(self.L.rtfhe_fft_plan_poly_mul, a, np.uint32, np.uint32, b)

    def get_twiddles(self):
        ifft, fft = np.empty(2 * self.N, np.float64), np.empty(2 * self.N, np.float64)
        self._ck(self.L.rtfhe_fft_plan_get_twiddles(self.h, _ptr(ifft), _ptr(fft)))
        return ifft, fft

    def set_twiddles(self, ifft_table, fft_table):
        a, b = _np(ifft_table, np.float64).reshape(2 * self.N), _np(fft_table, np.float64).reshape(2 * self.N)
        self._ck(self.L.rtfhe_fft_plan_set_twiddles(self.h, _ptr(a), _ptr(b)))


def device_link(dev_a, dev_b):
    """What the runtime reports between two devices of the node (rtfhe_device_link): {"can_access", "link", "link_type", "hops"}."""
    L = _ffi.load()
    can, lt, hops = C.c_int32(), C.c_uint32(), C.c_uint32()
    rc = L.rtfhe_device_link(dev_a, dev_b, C.byref(can), C.byref(lt), C.byref(hops))
    if rc != 0:
        raise RtfheError(rc, (L.rtfhe_last_error(None) or b"").decode())
    return {"can_access": can.value, "link_type": lt.value, "link": _ffi.PeerInfo.LINK_NAMES.get(lt.value, "type %d" % lt.value), "hops": hops.value}


def pinned_empty(shape, dtype=np.uint32):
    """numpy array over pinned host memory (rtfhe_host_alloc): host-pointer calls DMA straight from / into it.  The memory
    is released when the array (and every view of it) is garbage-collected."""
    L = _ffi.load()
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    ptr = L.rtfhe_host_alloc(n)
    if not ptr:
        raise RtfheError(_ffi.ERR_NOMEM, "rtfhe_host_alloc failed (needs a HIP device)")

    class _Owner:
        def __del__(self, _free=L.rtfhe_host_free, _p=ptr):
            _free(C.c_void_p(_p))
    buf = (C.c_char * n).from_address(ptr)
    arr = np.frombuffer(buf, dtype=dtype).reshape(shape)
    buf._owner = _Owner()
    return arr


# ---- host-side key generation / encryption (C ABI, no GPU needed) --------------------------------

def keygen(params, seed=None, want_bk=True, want_ksk=True):
    """seed None (production): every key bit, mask and noise sample comes from the OS CSPRNG.  An integer seed selects the
    TEST-ONLY deterministic generator (rtfhe_keygen_deterministic: reproducible, NOT secure)."""
    L = _ffi.load()
    key0 = np.empty(params.n, np.int32)
    key1 = np.empty(params.N, np.int32)
    bk = np.empty(params.bk_words, np.uint32) if want_bk else None
    ksk = np.empty(params.ksk_words, np.uint32) if want_ksk else None
    if seed is None:
        rc = L.rtfhe_keygen(C.byref(params), _ptr(key0), _ptr(key1), _ptr(bk), _ptr(ksk))
    else:
        rc = L.rtfhe_keygen_deterministic(C.byref(params), seed, _ptr(key0), _ptr(key1), _ptr(bk), _ptr(ksk))
    if rc != 0:
        raise RtfheError(rc, "rtfhe_keygen failed")
    return key0, key1, bk, ksk


def encrypt_bits(params, key0, bits, seed=None):
    """seed None (production): mask and noise from the OS CSPRNG; an integer seed = TEST-ONLY deterministic encryption."""
    L = _ffi.load()
    bits = _np(bits, np.uint8).reshape(-1)
    key0 = _np(key0, np.int32)
    out = np.empty((bits.size, params.n + 1), np.uint32)
    if seed is None:
        rc = L.rtfhe_tlwe_encrypt_bits(C.byref(params), _ptr(key0), _ptr(bits), _ptr(out), bits.size)
    else:
        rc = L.rtfhe_tlwe_encrypt_bits_deterministic(C.byref(params), _ptr(key0), seed, _ptr(bits), _ptr(out), bits.size)
    if rc != 0:
        raise RtfheError(rc, "rtfhe_tlwe_encrypt_bits failed")
    return out


def decrypt_bits(params, key0, cts):
    L = _ffi.load()
    cts = _np(cts, np.uint32).reshape(-1, params.n + 1)
    key0 = _np(key0, np.int32)
    bits = np.empty(cts.shape[0], np.uint8)
    rc = L.rtfhe_tlwe_decrypt_bits(C.byref(params), _ptr(key0), _ptr(cts), _ptr(bits), cts.shape[0])
    if rc != 0:
        raise RtfheError(rc, "rtfhe_tlwe_decrypt_bits failed")
    return bits


def phases(params, key0, cts):
    L = _ffi.load()
    cts = _np(cts, np.uint32).reshape(-1, params.n + 1)
    key0 = _np(key0, np.int32)
    ph = np.empty(cts.shape[0], np.uint32)
    rc = L.rtfhe_tlwe_phase(C.byref(params), _ptr(key0), _ptr(cts), _ptr(ph), cts.shape[0])
    if rc != 0:
        raise RtfheError(rc, "rtfhe_tlwe_phase failed")
    return ph


# ---- wire format (flat files; include/rtfhe.h) ----------------------------------------------------

def shard_range(count, d, n_dev):
    """[begin, end) of a count-gate host batch taken by entry d of an n_dev-device context (rtfhe_shard_range)."""
    L = _ffi.load()
    b, e = C.c_size_t(), C.c_size_t()
    rc = L.rtfhe_shard_range(count, d, n_dev, C.byref(b), C.byref(e))
    if rc != 0:
        raise RtfheError(rc, (L.rtfhe_last_error(None) or b"").decode())
    return b.value, e.value


def ksk_expand_ref(params, key0, key1, ksk, seed=None):
    """The reference's KeySwitchingKey shape u32[N][t][base][n+1] (hom_nand/src/tlwe.rs:243-245) from the compact key:
    entries t = 1 .. base-1 copied, entry t = base encrypted afresh (seed: TEST ONLY, deterministic)."""
    L = _ffi.load()
    key0, key1, ksk = _np(key0, np.int32), _np(key1, np.int32), _np(ksk, np.uint32).reshape(-1)
    base = 1 << params.ks_basebit
    out = np.empty(params.ksk_words // (base - 1) * base, np.uint32)
    if seed is None:
        rc = L.rtfhe_ksk_expand_ref(C.byref(params), _ptr(key0), _ptr(key1), _ptr(ksk), _ptr(out))
    else:
        rc = L.rtfhe_ksk_expand_ref_deterministic(C.byref(params), seed, _ptr(key0), _ptr(key1), _ptr(ksk), _ptr(out))
    if rc != 0:
        raise RuntimeError("rtfhe_ksk_expand_ref failed (%d)" % rc)
    return out.reshape(params.N, params.ks_t, base, params.n + 1)


def save_keys(path, params, key0=None, key1=None, bk=None, ksk=None):
    L = _ffi.load()
    arrs = [None if a is None else _np(a, dt) for a, dt in ((key0, np.int32), (key1, np.int32), (bk, np.uint32), (ksk, np.uint32))]
    rc = L.rtfhe_keys_write(path.encode(), C.byref(params), *[_ptr(a) for a in arrs])
    if rc != 0:
        raise RtfheError(rc, "rtfhe_keys_write failed")


def load_keys(path, want_bk=True, want_ksk=True):
    """Returns (params, key0, key1, bk, ksk); sections absent from the file (or not wanted) come back as None."""
    L = _ffi.load()
    p, flags = Params(), C.c_uint32()
    if L.rtfhe_keys_read_header(path.encode(), C.byref(p), C.byref(flags)) != 0:
        raise RtfheError(ERR_INVALID, "not an rtfhe key file: " + path)
    f = flags.value
    key0 = np.empty(p.n, np.int32) if f & 4 else None
    key1 = np.empty(p.N, np.int32) if f & 4 else None
    bk = np.empty(p.bk_words, np.uint32) if (f & 1 and want_bk) else None
    ksk = np.empty(p.ksk_words, np.uint32) if (f & 2 and want_ksk) else None
    if L.rtfhe_keys_read(path.encode(), _ptr(key0), _ptr(key1), _ptr(bk), _ptr(ksk)) != 0:
        raise RtfheError(ERR_INVALID, "corrupt rtfhe key file (checksum or length): " + path)
    return p, key0, key1, bk, ksk


def save_tlwe(path, params, cts):
    L = _ffi.load()
    cts = _np(cts, np.uint32).reshape(-1, params.n + 1)
    if L.rtfhe_tlwe_write(path.encode(), params.n, _ptr(cts), cts.shape[0]) != 0:
        raise RtfheError(ERR_INVALID, "rtfhe_tlwe_write failed")


def load_tlwe(path):
    L = _ffi.load()
    n, count = C.c_int32(), C.c_uint64()
    if L.rtfhe_tlwe_read(path.encode(), C.byref(n), C.byref(count), None, 0) != 0:
        raise RtfheError(ERR_INVALID, "not an rtfhe ciphertext file: " + path)
    out = np.empty((count.value, n.value + 1), np.uint32)
    if L.rtfhe_tlwe_read(path.encode(), C.byref(n), C.byref(count), _ptr(out), count.value) != 0:
        raise RtfheError(ERR_INVALID, "corrupt rtfhe ciphertext file: " + path)
    return out
