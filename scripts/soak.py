#!/usr/bin/env python3
"""Soak run: the same batch launched many times must give the same words every time (the kernels' pair-only flag synchronisation and split
first stage have no fences to hide a race behind: a rare ordering bug would show as a rare differing output).  Prints one JSON line per
(N, backend, batch size): launches, mismatching launches.  usage: soak.py [launches]"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import rustfhe_amd as R

DEFAULT_SHAPES = ((1024, ("fft", "ntt", "xfft"), (1024, 768, 512, 300, 1280, 1500)), (2048, ("fft", "ntt", "xfft"), (1024, 768, 512, 100)))


def run(launches=200, shapes=DEFAULT_SHAPES, emit=print):
    """Returns the number of failures (launches whose words differ from the first launch's + batches that do not decrypt)."""
    st = torch.cuda.current_stream().cuda_stream
    bad_total = 0
    for N, backends, counts in shapes:
        P = R.Params(N=N)
        key0, key1, bk, ksk = R.keygen(P, 7 + N)
        e = R.Engine(P, 0)
        e.load_bk_torus(bk); e.load_ksk(ksk)
        rng = np.random.default_rng(N)
        G = max(counts)
        b0, b1 = rng.integers(0, 2, G).astype(np.uint8), rng.integers(0, 2, G).astype(np.uint8)
        d0 = torch.from_numpy(R.encrypt_bits(P, key0, b0, 1).view(np.int32)).cuda()
        d1 = torch.from_numpy(R.encrypt_bits(P, key0, b1, 2).view(np.int32)).cuda()
        for be in backends:
            e.set_backend({"fft": 0, "ntt": 1, "xfft": 2}[be])
            for G in counts:
                ref = torch.empty_like(d0[:G]); out = torch.empty_like(d0[:G])
                e.gate_batch_dev(R.NAND, d0, d1, ref, G, st)
                torch.cuda.synchronize()
                dec = R.decrypt_bits(P, key0, ref.cpu().numpy().view(np.uint32))
                ok_dec = bool(np.array_equal(np.asarray(dec, np.uint8), 1 - (b0[:G] & b1[:G])))
                n_launch = launches if be != "ntt" else max(20, launches // 4)
                bad = 0
                for _ in range(n_launch):
                    out.zero_()
                    e.gate_batch_dev(R.NAND, d0, d1, out, G, st)
                    if not torch.equal(out, ref):
                        bad += 1
                bad_total += bad + (0 if ok_dec else 1)
                emit(json.dumps({"N": N, "backend": be, "gates": G, "launches": n_launch, "mismatching_launches": bad, "decrypts": ok_dec}))
        e.close()
    return bad_total


if __name__ == "__main__":
    bad_total = run(int(sys.argv[1]) if len(sys.argv) > 1 else 200, emit=lambda s: print(s, flush=True))
    print("soak:", "clean" if bad_total == 0 else "%d FAILURES" % bad_total)
    sys.exit(0 if bad_total == 0 else 1)
