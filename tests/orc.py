"""ctypes bindings for the CPU oracle (oracle/liboracle.so) and, when it has been built, the
reference's own compiled spqlios (oracle/_ref/libspqlios_ref.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (rustfhe_amd/) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "liboracle.so")
REF_PATH = os.path.join(ORACLE_DIR, "_ref", "libspqlios_ref.so")

NAND, AND, OR, XOR, NOT, COPY, ANDNY = range(7)
BACKEND_MIRROR, BACKEND_EXACT, BACKEND_HOOK = 0, 1, 2


class Params(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ("n", "N", "nbit", "l", "bgbit", "ks_t", "ks_basebit")]

    def __init__(self, n=635, N=1024, nbit=None, l=3, bgbit=6, ks_t=8, ks_basebit=2):
        super().__init__()
        self.n, self.N, self.l, self.bgbit, self.ks_t, self.ks_basebit = n, N, l, bgbit, ks_t, ks_basebit
        self.nbit = nbit if nbit is not None else int(N).bit_length() - 1

    @property
    def trgsw_words(self):
        return 2 * 2 * self.l * self.N

    @property
    def ksk_words(self):
        return self.N * self.ks_t * ((1 << self.ks_basebit) - 1) * (self.n + 1)


class Rng(C.Structure):
    _fields_ = [("s", C.c_uint64 * 4)]


def build(force=False):
    src = [os.path.join(ORACLE_DIR, f) for f in ("tfhe_oracle.c", "tfhe_oracle.h")]
    stale = (not os.path.exists(LIB_PATH)) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in src)
    if force or stale:
        subprocess.check_call(["make", "-C", ORACLE_DIR, "all"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference/utils/src/spqlios") and not os.path.exists(REF_PATH):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "ref"], stdout=subprocess.DEVNULL)


_lib = None


def _p(a, ct):
    if a is None:
        return None
    return a.ctypes.data_as(C.POINTER(ct))


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(LIB_PATH)
    vp, i32, u32, f64p = C.c_void_p, C.c_int32, C.c_uint32, C.POINTER(C.c_double)
    u32p, i32p, PP = C.POINTER(C.c_uint32), C.POINTER(C.c_int32), C.POINTER(Params)
    sig = {
        "orc_plan_new": (vp, [i32]),
        "orc_plan_free": (None, [vp]),
        "orc_plan_set_backend": (None, [vp, C.c_int]),
        "orc_plan_set_hooks": (None, [vp, vp, vp, vp]),
        "orc_plan_export_tables": (None, [vp, f64p, f64p]),
        "orc_plan_import_tables": (None, [vp, f64p, f64p]),
        "orc_ifft_i32": (None, [vp, f64p, i32p]),
        "orc_ifft_f64": (None, [vp, f64p, f64p]),
        "orc_fft_u32": (None, [vp, u32p, f64p]),
        "orc_fft_f64": (None, [vp, f64p, f64p]),
        "orc_poly_mul": (None, [vp, u32p, u32p, u32p]),
        "orc_hadamard": (None, [i32, f64p, f64p, f64p]),
        "orc_torus_from_f32": (u32, [C.c_float]),
        "orc_make_decomp_mask": (u32, [u32, u32]),
        "orc_inline_decomp_mask": (u32, [u32, u32]),
        "orc_decomp_scalar": (None, [u32, u32, u32, i32, i32p]),
        "orc_decomp_u32_scalar": (None, [u32, u32, i32, u32p]),
        "orc_decomp_poly": (None, [i32, u32p, u32, u32, i32, i32p]),
        "orc_rotate_u32": (None, [i32, u32p, i32, u32p]),
        "orc_rotate_i32": (None, [i32, i32p, i32, i32p]),
        "orc_negacyclic_mul_u32": (None, [i32, u32p, i32p, u32p]),
        "orc_trgsw_to_fft": (None, [vp, u32p, f64p, C.c_size_t]),
        "orc_external_product": (None, [PP, vp, f64p, u32p, u32p, u32p]),
        "orc_cmux": (None, [PP, vp, f64p, u32p, u32p, u32p, u32p]),
        "orc_blind_rotate": (None, [PP, vp, f64p, u32p, u32p, i32, u32p]),
        "orc_sample_extract": (None, [i32, u32p, i32, u32p]),
        "orc_key_switch": (None, [PP, u32p, u32p, u32p]),
        "orc_key_switch_ref": (None, [PP, u32p, u32p, u32p]),
        "orc_gate_linear": (None, [PP, C.c_int, u32p, u32p, u32p]),
        "orc_bootstrap": (None, [PP, vp, f64p, u32p, u32p, u32p, u32p]),
        "orc_gate": (None, [PP, vp, C.c_int, f64p, u32p, u32p, u32p, u32p, u32p]),
        "orc_mux": (None, [PP, vp, f64p, u32p, u32p, u32p, u32p, u32p, u32p]),
        "orc_gate_batch_mt": (C.c_double, [PP, C.c_int, C.c_int, f64p, u32p, u32p, u32p, u32p, u32p, C.c_size_t, C.c_int]),
        "orc_gate_batch_mt_numa": (C.c_double, [PP, C.c_int, C.c_int, f64p, u32p, u32p, u32p, C.c_size_t, u32p, C.c_size_t, C.c_int, i32p, i32p, C.c_int]),
        "orc_set_mt_hooks": (None, [vp, vp, vp]),
        "orc_rng_seed": (None, [C.POINTER(Rng), C.c_uint64]),
        "orc_rng_next": (C.c_uint64, [C.POINTER(Rng)]),
        "orc_rng_uniform_torus": (u32, [C.POINTER(Rng)]),
        "orc_rng_gaussian_torus": (u32, [C.POINTER(Rng), C.c_float]),
        "orc_gen_binary_key": (None, [C.POINTER(Rng), i32, i32p]),
        "orc_tlwe_encrypt": (None, [C.POINTER(Rng), i32, i32p, u32, C.c_float, u32p]),
        "orc_tlwe_phase": (u32, [i32, i32p, u32p]),
        "orc_torus2binary": (C.c_int, [u32]),
        "orc_binary2torus": (u32, [C.c_int]),
        "orc_trlwe_encrypt": (None, [C.POINTER(Rng), vp, i32, i32p, u32p, C.c_float, u32p]),
        "orc_trlwe_phase": (None, [vp, i32, i32p, u32p, u32p]),
        "orc_trgsw_encrypt": (None, [C.POINTER(Rng), vp, PP, i32p, i32, C.c_float, u32p]),
        "orc_bk_gen": (None, [C.POINTER(Rng), vp, PP, i32p, i32p, C.c_float, u32p]),
        "orc_ksk_gen": (None, [C.POINTER(Rng), PP, i32p, i32p, C.c_float, u32p]),
        "orc_ksk_expand_ref": (None, [C.POINTER(Rng), PP, i32p, i32p, C.c_float, u32p, u32p]),
        "orc_fnv64": (C.c_uint64, [vp, C.c_size_t]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    _lib = L
    return L


def have_ref():
    build()
    return os.path.exists(REF_PATH)


def ref_lib():
    """The reference's own compiled spqlios.  One N per process (function-local static 2/N)."""
    R = C.CDLL(REF_PATH)
    R.Spqlios_new.restype = C.c_void_p
    R.Spqlios_new.argtypes = [C.c_int32]
    for f in ("Spqlios_ifft_i32", "Spqlios_ifft_u32", "Spqlios_fft_u32", "Spqlios_ifft", "Spqlios_fft"):
        getattr(R, f).restype = None
        getattr(R, f).argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    R.Spqlios_poly_mul.restype = None
    R.Spqlios_poly_mul.argtypes = [C.c_void_p] * 4
    for f in ("new_fft_table", "new_ifft_table"):
        getattr(R, f).restype = C.c_void_p
        getattr(R, f).argtypes = [C.c_int32]
    return R


class Plan:
    """orc_plan wrapper; numpy in / numpy out."""

    def __init__(self, N, backend=BACKEND_MIRROR):
        self.L = lib()
        self.N = N
        self.h = self.L.orc_plan_new(N)
        assert self.h, "bad N"
        self.L.orc_plan_set_backend(self.h, backend)
        self._keep = None

    def use_reference_fft(self):
        """Route forward/inverse transforms through the reference's compiled AVX spqlios."""
        R = ref_lib()
        handle = R.Spqlios_new(self.N)
        fwd = C.cast(R.Spqlios_ifft_i32, C.c_void_p)
        inv = C.cast(R.Spqlios_fft_u32, C.c_void_p)
        self.L.orc_plan_set_hooks(self.h, handle, fwd, inv)
        self._keep = (R, handle)
        return self

    def __del__(self):
        try:
            self.L.orc_plan_free(self.h)
        except Exception:
            pass

    def tables(self):
        a = np.zeros(2 * self.N, np.float64)
        b = np.zeros(2 * self.N, np.float64)
        self.L.orc_plan_export_tables(self.h, _p(a, C.c_double), _p(b, C.c_double))
        return a, b

    def ifft_i32(self, src):
        src = np.ascontiguousarray(src, np.int32)
        res = np.empty(self.N, np.float64)
        self.L.orc_ifft_i32(self.h, _p(res, C.c_double), _p(src, C.c_int32))
        return res

    def fft_u32(self, src):
        src = np.ascontiguousarray(src, np.float64)
        res = np.empty(self.N, np.uint32)
        self.L.orc_fft_u32(self.h, _p(res, C.c_uint32), _p(src, C.c_double))
        return res

    def ifft_f64(self, src):
        src = np.ascontiguousarray(src, np.float64)
        res = np.empty(self.N, np.float64)
        self.L.orc_ifft_f64(self.h, _p(res, C.c_double), _p(src, C.c_double))
        return res

    def fft_f64(self, src):
        src = np.ascontiguousarray(src, np.float64)
        res = np.empty(self.N, np.float64)
        self.L.orc_fft_f64(self.h, _p(res, C.c_double), _p(src, C.c_double))
        return res

    def poly_mul(self, a, b):
        a = np.ascontiguousarray(a, np.uint32)
        b = np.ascontiguousarray(b, np.uint32)
        res = np.empty(self.N, np.uint32)
        self.L.orc_poly_mul(self.h, _p(res, C.c_uint32), _p(a, C.c_uint32), _p(b, C.c_uint32))
        return res


def rotate(p, n):
    p = np.ascontiguousarray(p)
    out = np.empty_like(p)
    if p.dtype == np.int32:
        lib().orc_rotate_i32(len(p), _p(p, C.c_int32), n, _p(out, C.c_int32))
    else:
        p = p.astype(np.uint32)
        out = np.empty_like(p)
        lib().orc_rotate_u32(len(p), _p(p, C.c_uint32), n, _p(out, C.c_uint32))
    return out


def decomp_scalar(x, bits, mask, l):
    out = np.empty(l, np.int32)
    lib().orc_decomp_scalar(x, bits, mask, l, _p(out, C.c_int32))
    return out.tolist()


def decomp_u32_scalar(x, bits, l):
    out = np.empty(l, np.uint32)
    lib().orc_decomp_u32_scalar(x, bits, l, _p(out, C.c_uint32))
    return out.tolist()


def negacyclic_mul(a, b):
    a = np.ascontiguousarray(a, np.uint32)
    b = np.ascontiguousarray(b, np.int32)
    res = np.empty(len(a), np.uint32)
    lib().orc_negacyclic_mul_u32(len(a), _p(a, C.c_uint32), _p(b, C.c_int32), _p(res, C.c_uint32))
    return res


class Keys:
    """Deterministic synthetic key set (own seeded RNG; the reference's thread_rng is unseedable)."""

    def __init__(self, params, seed, plan=None, alpha_bk=2.0 ** -25, alpha_ks=2.0 ** -15, fft=True):
        L = lib()
        self.p = params
        self.seed = seed
        plan = plan or Plan(params.N)
        rng = Rng()
        L.orc_rng_seed(C.byref(rng), seed)
        self.key0 = np.empty(params.n, np.int32)
        self.key1 = np.empty(params.N, np.int32)
        L.orc_gen_binary_key(C.byref(rng), params.n, _p(self.key0, C.c_int32))
        L.orc_gen_binary_key(C.byref(rng), params.N, _p(self.key1, C.c_int32))
        self.bk_t = np.empty(params.n * params.trgsw_words, np.uint32)
        L.orc_bk_gen(C.byref(rng), plan.h, C.byref(params), _p(self.key0, C.c_int32), _p(self.key1, C.c_int32),
                     C.c_float(alpha_bk), _p(self.bk_t, C.c_uint32))
        self.ksk = np.empty(params.ksk_words, np.uint32)
        L.orc_ksk_gen(C.byref(rng), C.byref(params), _p(self.key1, C.c_int32), _p(self.key0, C.c_int32),
                      C.c_float(alpha_ks), _p(self.ksk, C.c_uint32))
        self.bk_f = None
        if fft:
            self.bk_f = np.empty(params.n * params.trgsw_words, np.float64)
            L.orc_trgsw_to_fft(plan.h, _p(self.bk_t, C.c_uint32), _p(self.bk_f, C.c_double),
                               params.n * 2 * 2 * params.l)
        self.rng = rng

    def ksk_ref(self, seed=0x4b534b, alpha_ks=2.0 ** -15):
        """The key-switching key in the reference's own container shape u32[N][t][base][n+1] (tlwe.rs:243-245): this key set's
        rows plus the never-read entry t = base of every level, encrypted with a generator of its own."""
        L = lib()
        rng = Rng()
        L.orc_rng_seed(C.byref(rng), seed)
        base = 1 << self.p.ks_basebit
        out = np.empty(self.p.ksk_words // (base - 1) * base, np.uint32)
        L.orc_ksk_expand_ref(C.byref(rng), C.byref(self.p), _p(self.key1, C.c_int32), _p(self.key0, C.c_int32),
                             C.c_float(alpha_ks), _p(self.ksk, C.c_uint32), _p(out, C.c_uint32))
        return out

    def encrypt_bits(self, bits, alpha=2.0 ** -15, rng=None):
        L = lib()
        rng = rng or self.rng
        n = self.p.n
        out = np.empty((len(bits), n + 1), np.uint32)
        for g, b in enumerate(bits):
            L.orc_tlwe_encrypt(C.byref(rng), n, _p(self.key0, C.c_int32), L.orc_binary2torus(int(b)),
                               C.c_float(alpha), _p(out[g], C.c_uint32))
        return out

    def phase(self, ct):
        return lib().orc_tlwe_phase(self.p.n, _p(self.key0, C.c_int32), _p(np.ascontiguousarray(ct, np.uint32), C.c_uint32))

    def decrypt_bits(self, cts):
        L = lib()
        return [L.orc_torus2binary(self.phase(ct)) for ct in np.ascontiguousarray(cts, np.uint32)]


def external_product(params, plan, trgsw_f, trgsw_t, trlwe):
    out = np.empty(2 * params.N, np.uint32)
    lib().orc_external_product(C.byref(params), plan.h, _p(trgsw_f, C.c_double), _p(trgsw_t, C.c_uint32),
                               _p(np.ascontiguousarray(trlwe, np.uint32), C.c_uint32), _p(out, C.c_uint32))
    return out


def blind_rotate(params, plan, bk_f, bk_t, tlwe, steps=None):
    acc = np.empty(2 * params.N, np.uint32)
    lib().orc_blind_rotate(C.byref(params), plan.h, _p(bk_f, C.c_double), _p(bk_t, C.c_uint32),
                           _p(np.ascontiguousarray(tlwe, np.uint32), C.c_uint32),
                           params.n if steps is None else steps, _p(acc, C.c_uint32))
    return acc


def sample_extract(params, trlwe, index=0):
    out = np.empty(params.N + 1, np.uint32)
    lib().orc_sample_extract(params.N, _p(np.ascontiguousarray(trlwe, np.uint32), C.c_uint32), index, _p(out, C.c_uint32))
    return out


def key_switch(params, ksk, tlwe1):
    out = np.empty(params.n + 1, np.uint32)
    lib().orc_key_switch(C.byref(params), _p(ksk, C.c_uint32), _p(np.ascontiguousarray(tlwe1, np.uint32), C.c_uint32),
                         _p(out, C.c_uint32))
    return out


def key_switch_ref(params, ksk_ref, tlwe1):
    out = np.empty(params.n + 1, np.uint32)
    lib().orc_key_switch_ref(C.byref(params), _p(ksk_ref, C.c_uint32), _p(np.ascontiguousarray(tlwe1, np.uint32), C.c_uint32),
                             _p(out, C.c_uint32))
    return out


def gate_linear(params, op, in0, in1):
    out = np.empty(params.n + 1, np.uint32)
    in1 = in0 if in1 is None else in1
    lib().orc_gate_linear(C.byref(params), op, _p(np.ascontiguousarray(in0, np.uint32), C.c_uint32),
                          _p(np.ascontiguousarray(in1, np.uint32), C.c_uint32), _p(out, C.c_uint32))
    return out


def gate(params, plan, op, bk_f, bk_t, ksk, in0, in1):
    out = np.empty(params.n + 1, np.uint32)
    in1 = in0 if in1 is None else in1
    lib().orc_gate(C.byref(params), plan.h, op, _p(bk_f, C.c_double), _p(bk_t, C.c_uint32), _p(ksk, C.c_uint32),
                   _p(np.ascontiguousarray(in0, np.uint32), C.c_uint32),
                   _p(np.ascontiguousarray(in1, np.uint32), C.c_uint32), _p(out, C.c_uint32))
    return out


def mux(params, plan, bk_f, bk_t, ksk, c, in0, in1):
    out = np.empty(params.n + 1, np.uint32)
    lib().orc_mux(C.byref(params), plan.h, _p(bk_f, C.c_double), _p(bk_t, C.c_uint32), _p(ksk, C.c_uint32),
                  _p(np.ascontiguousarray(c, np.uint32), C.c_uint32),
                  _p(np.ascontiguousarray(in0, np.uint32), C.c_uint32),
                  _p(np.ascontiguousarray(in1, np.uint32), C.c_uint32), _p(out, C.c_uint32))
    return out


def gate_batch_mt(params, op, bk_f, bk_t, ksk, in0, in1, nthreads, backend=BACKEND_MIRROR):
    in0 = np.ascontiguousarray(in0, np.uint32)
    in1 = in0 if in1 is None else np.ascontiguousarray(in1, np.uint32)
    out = np.empty_like(in0)
    secs = lib().orc_gate_batch_mt(C.byref(params), backend, op, _p(bk_f, C.c_double), _p(bk_t, C.c_uint32),
                                   _p(ksk, C.c_uint32), _p(in0, C.c_uint32), _p(in1, C.c_uint32),
                                   _p(out, C.c_uint32), in0.shape[0], nthreads)
    return out, secs


def host_topology():
    """What the all-core baseline needs to know about the host: the hardware threads this process may run on, which physical core and
    memory node each belongs to, and the CPU-time quota of the enclosing cgroup (a container may see every CPU of the machine and still be
    entitled to a few cores' worth of time).  Plain reads of /proc and /sys; anything unreadable degrades to 'one node, no quota'."""
    threads = sorted(os.sched_getaffinity(0))
    core_of, node_of = {}, {}
    try:
        cur = {}
        with open("/proc/cpuinfo") as f:
            for ln in f.read().split("\n") + [""]:
                if ":" in ln:
                    k, v = ln.split(":", 1)
                    cur[k.strip()] = v.strip()
                elif cur:
                    cpu = int(cur.get("processor", -1))
                    core_of[cpu] = (cur.get("physical id", "0"), cur.get("core id", str(cpu)))
                    cur = {}
    except OSError:
        pass
    try:
        base = "/sys/devices/system/node"
        for d in sorted(os.listdir(base)):
            if d.startswith("node") and d[4:].isdigit():
                with open(os.path.join(base, d, "cpulist")) as f:
                    for part in f.read().strip().split(","):
                        if part:
                            lo, _, hi = part.partition("-")
                            for cpu in range(int(lo), int(hi or lo) + 1):
                                node_of[cpu] = int(d[4:])
    except (OSError, ValueError):
        pass
    quota = None
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: None if t.split()[0] == "max" else float(t.split()[0]) / float(t.split()[1])),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", None)):
        try:
            with open(path) as f:
                t = f.read().strip()
            if parse:
                quota = parse(t)
            elif int(t) > 0:
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f2:
                    quota = int(t) / float(f2.read().strip())
            break
        except (OSError, ValueError, IndexError, ZeroDivisionError):
            continue
    if os.environ.get("RTFHE_TEST_CPU_QUOTA"):          # tests: pretend the cgroup grants this many cores
        quota = float(os.environ["RTFHE_TEST_CPU_QUOTA"])
    first_of_core = {}
    for cpu in threads:
        first_of_core.setdefault(core_of.get(cpu, ("0", str(cpu))), cpu)
    model = ""
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name"):
                    model = ln.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    core_ids = {k: i for i, k in enumerate(sorted(first_of_core))}
    return {"hw_threads": threads, "one_thread_per_core": sorted(first_of_core.values()), "node_of": {c: node_of.get(c, 0) for c in threads},
            "core_of": {c: core_ids[core_of.get(c, ("0", str(c)))] for c in threads},
            "nodes": sorted(set(node_of.get(c, 0) for c in threads)), "cgroup_cpu_quota": quota, "cpu_model": model}


def gate_batch_mt_numa(params, op, bk_f, ksk, in0, in1, count, cpus, node_of, backend=BACKEND_MIRROR, pin=True):
    """`count` gates over the inputs in0 / in1 (gate g takes input g % len(in0)) on len(cpus) threads, thread t pinned to cpus[t] (pin) and
    reading the key replica of memory node node_of[cpus[t]] (first-touched by a thread of that node).  Returns (out[count], seconds)."""
    in0 = np.ascontiguousarray(in0, np.uint32)
    in1 = in0 if in1 is None else np.ascontiguousarray(in1, np.uint32)
    out = np.empty((count, in0.shape[1]), np.uint32)
    nodes = sorted(set(node_of[c] for c in cpus))
    cpu_arr = np.array([c if pin else -1 for c in cpus], np.int32)
    node_arr = np.array([nodes.index(node_of[c]) for c in cpus], np.int32)
    secs = lib().orc_gate_batch_mt_numa(C.byref(params), backend, op, _p(bk_f, C.c_double), _p(ksk, C.c_uint32), _p(in0, C.c_uint32),
                                        _p(in1, C.c_uint32), in0.shape[0], _p(out, C.c_uint32), count, len(cpus),
                                        _p(cpu_arr, C.c_int32), _p(node_arr, C.c_int32), len(nodes))
    if secs < 0:
        raise MemoryError("orc_gate_batch_mt_numa: key replica allocation failed")
    return out, secs


_mt_ref = None


def use_reference_fft_in_mt():
    """Worker threads of gate_batch_mt(backend=BACKEND_HOOK) each get their own handle of the reference's compiled spqlios."""
    global _mt_ref
    if _mt_ref is None:
        R = ref_lib()
        lib().orc_set_mt_hooks(C.cast(R.Spqlios_new, C.c_void_p), C.cast(R.Spqlios_ifft_i32, C.c_void_p), C.cast(R.Spqlios_fft_u32, C.c_void_p))
        _mt_ref = R


def fnv64(arr):
    arr = np.ascontiguousarray(arr)
    return lib().orc_fnv64(arr.ctypes.data_as(C.c_void_p), arr.nbytes)
