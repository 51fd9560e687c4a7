"""CPU, world_size 2, gloo: the scatter -> per-rank bootstrap -> gather flow of rustfhe_amd/shard.py.  The per-rank
compute here is the CPU oracle on a small parameter set (tests may use the oracle as a stand-in checker; on a GPU
node the callback is shard.engine_compute)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, count, q, op_name="NAND"):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import orc
    from rustfhe_amd.shard import ShardedGates, partition
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    P = orc.Params(n=5)
    K = orc.Keys(P, 321)            # same seed on every rank = replicated keys
    pl = orc.Plan(P.N)

    def compute(op, a, b):
        a = a.numpy().view(np.uint32)
        b = a if b is None else b.numpy().view(np.uint32)
        out = np.stack([orc.gate(P, pl, op, K.bk_f, None, K.ksk, x, y) for x, y in zip(a, b)])
        return torch.from_numpy(out.view(np.int32))

    sg = ShardedGates(compute, P.n + 1, torch.device("cpu"))
    in0 = in1 = None
    bits = None
    if rank == 0:
        rng = np.random.default_rng(7)
        bits = (rng.integers(0, 2, count), rng.integers(0, 2, count))
        in0 = torch.from_numpy(K.encrypt_bits(bits[0]).view(np.int32))
        in1 = torch.from_numpy(K.encrypt_bits(bits[1]).view(np.int32))
    op = getattr(orc, op_name)
    unary = op_name in ("NOT", "COPY")
    out = sg.run(op, in0, None if unary else in1, count, sync=(lambda: None))
    assert set(sg.last_timing) == {"scatter_s", "compute_s", "gather_s"}
    res = None
    if rank == 0:
        o = out.numpy().view(np.uint32)
        a0 = in0.numpy().view(np.uint32)
        exp = np.stack([orc.gate(P, pl, op, K.bk_f, None, K.ksk, x, y) for x, y in zip(a0, a0 if unary else in1.numpy().view(np.uint32))]) \
            if count else np.empty((0, P.n + 1), np.uint32)
        want = {"NAND": 1 - (bits[0] & bits[1]), "NOT": 1 - bits[0], "COPY": bits[0]}[op_name]
        res = (bool(np.array_equal(o, exp)), K.decrypt_bits(o) == list(want), partition(count, world))
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        q.put(res)


@pytest.mark.parametrize("count,op_name", [(7, "NAND"), (2, "NAND"), (1, "NAND"), (5, "NOT")])
def test_scatter_bootstrap_gather_world2(count, op_name):
    """op NOT: a unary gate has no second input -- every rank must learn that from the root (it used to deadlock)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, count, q, op_name)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    same, dec_ok, parts = res
    assert same and dec_ok
    assert parts[0][0] == 0 and parts[-1][1] == count and all(a[1] == b[0] for a, b in zip(parts, parts[1:]))


@pytest.mark.parametrize("count", [7, 2, 0])
def test_scatter_bootstrap_gather_world3_ragged_and_empty_ranks(count):
    """World size 3: 7 gates = 2 + 2 + 3 (ragged); 2 gates = 0 + 1 + 1 -- the ROOT itself gets no gate and still scatters and gathers;
    0 gates: nobody computes, everybody returns.  The root posts its sends, launches its own shard and only then posts the receives of the results (shard.py: run explains the order)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 3, port, count, q, "NAND")) for r in range(3)]
    for p in procs:
        p.start()
    res = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    same, dec_ok, parts = res
    assert same and dec_ok
    sizes = [e - b for b, e in parts]
    assert sum(sizes) == count and max(sizes) - min(sizes) <= 1
    if count == 2:
        assert sizes == [0, 1, 1]


def test_partition_properties():
    from rustfhe_amd.shard import partition
    for count in (0, 1, 5, 1024, 65536, 65537):
        for world in (1, 2, 3, 8):
            p = partition(count, world)
            sizes = [e - b for b, e in p]
            assert sum(sizes) == count and max(sizes) - min(sizes) <= 1 and p[0][0] == 0 and p[-1][1] == count
