#!/usr/bin/env python3
"""Host-pointer rate vs resident-buffer rate at 1024 gates: rtfhe_gate_batch from pageable memory (staged
through the context's pinned buffers, or handed to hipMemcpyAsync as is with RTFHE_STAGING=0), from rtfhe_host_alloc memory,
and rtfhe_gate_batch_dev on resident tensors."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch
    import rustfhe_amd as R
    p = R.Params()
    key0, key1, bk, ksk = R.keygen(p, 20211003)
    e = R.Engine(p, 0)
    e.load_bk_torus(bk)
    e.load_ksk(ksk)
    G = 1024
    rng = np.random.default_rng(3)
    b0, b1 = rng.integers(0, 2, G).astype(np.uint8), rng.integers(0, 2, G).astype(np.uint8)
    c0, c1 = R.encrypt_bits(p, key0, b0, 1), R.encrypt_bits(p, key0, b1, 2)
    p0, p1 = R.pinned_empty(c0.shape), R.pinned_empty(c1.shape)
    p0[:], p1[:] = c0, c1
    d0, d1 = torch.from_numpy(c0.view(np.int32)).cuda(), torch.from_numpy(c1.view(np.int32)).cuda()
    dout = torch.empty_like(d0)
    st = torch.cuda.current_stream().cuda_stream

    def t(fn, reps=8):
        fn(); fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3
    dev = t(lambda: e.gate_batch_dev(R.NAND, d0, d1, dout, G, st))
    pageable = t(lambda: e.gate_batch(R.NAND, c0, c1))
    pinned = t(lambda: e.gate_batch(R.NAND, p0, p1))
    mux = t(lambda: e.mux_batch(c0, c1, c0), reps=3)
    print("RTFHE_STAGING=%s  1024 gates: resident %.3f ms | host pageable %.3f ms (+%.1f%%) | host pinned (rtfhe_host_alloc) %.3f ms (+%.1f%%) | mux %.3f ms (= %.2f x gate)"
          % (os.environ.get("RTFHE_STAGING", "1"), dev, pageable, 100 * (pageable / dev - 1), pinned, 100 * (pinned / dev - 1), mux, mux / dev))
    assert np.array_equal(e.gate_batch(R.NAND, c0, c1), dout.cpu().numpy().view(np.uint32))
    e.close()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        main()
    else:   # two fresh processes (the knob is read at context creation); children are started before this one touches the GPU
        for v in ("1", "0"):
            subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, RTFHE_STAGING=v), check=False)
