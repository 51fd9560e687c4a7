"""rustfhe_amd -- MI355X (gfx950) engine for the HomNAND hot path of hideki1217/rusTfhe.

The product is the C-ABI shared library librtfhe_hip.so (include/rtfhe.h, rustfhe_amd/csrc/); this
package is the thin Python host side above it.  There is no CPU fallback anywhere in the package.
"""
from ._ffi import AND, ANDNY, COPY, NAND, NOT, OR, XOR, Params, load  # noqa: F401
from .engine import (Engine, FftPlan, RtfheError, decrypt_bits, device_link, encrypt_bits, keygen, ksk_expand_ref, load_keys, load_tlwe, phases, pinned_empty,  # noqa: F401
                     save_keys, save_tlwe, shard_range)

__all__ = ["Engine", "FftPlan", "Params", "RtfheError", "keygen", "ksk_expand_ref", "encrypt_bits", "decrypt_bits", "phases", "save_keys", "load_keys", "save_tlwe", "load_tlwe", "pinned_empty", "shard_range", "device_link",
           "NAND", "AND", "OR", "XOR", "NOT", "COPY", "ANDNY", "load"]
