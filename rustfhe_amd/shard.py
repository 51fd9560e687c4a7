"""Sharding of a gate batch over the GPUs of one node.

Gates are independent: rank g bootstraps the contiguous range [g*B/G, (g+1)*B/G) with its own replica of the
keys.  The only communication is the scatter of input ciphertexts from the root and the gather of results back:
point-to-point sends/receives of *views* of the root's tensors (RCCL over xGMI when the process group's backend is
"nccl" -- RCCL has no native scatter/gather, a grouped send/recv is what one is -- and the same code on "gloo" for
CPU tests).  There is no reduction and no exchange inside the path; nothing is padded or staged on the root.
"""
import torch
import torch.distributed as dist


def partition(count, world):
    """Contiguous ranges, sizes differ by at most one."""
    return [(count * g // world, count * (g + 1) // world) for g in range(world)]


class ShardedGates:
    """compute(op, in0, in1) -> out works on this rank's shard (tensors int32 [k, n+1] on `device`, the device the
    process group communicates on: the rank's GPU for nccl, the CPU for gloo)."""

    def __init__(self, compute, width, device, group=None, root=0):
        self.compute, self.width, self.device, self.group, self.root = compute, width, device, group, root
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.last_timing = None

    def _peer(self, r):
        return dist.get_global_rank(self.group, r) if self.group is not None else r

    def _header(self, count, has_in1):
        """count and the arity travel from the root: every rank takes the same branches (a unary gate has no in1)."""
        h = torch.tensor([count, 1 if has_in1 else 0] if self.rank == self.root else [0, 0], dtype=torch.int64, device=self.device)
        if self.world > 1:
            dist.broadcast(h, src=self._peer(self.root), group=self.group)
        return int(h[0].item()), bool(h[1].item())

    def _post_scatter(self, fulls, parts):
        """Root: posts the sends of every other rank's rows (views of the full tensors: nothing is staged) and returns (works, own rows)
        WITHOUT waiting -- the root's own shard does not depend on them.  Other ranks: receive their rows and wait."""
        lo, hi = parts[self.rank]
        if self.rank == self.root:
            ops, mine = [], []
            for f in fulls:
                assert f.dtype == torch.int32 and f.is_contiguous() and f.device == torch.device(self.device)
                for r, (b, e) in enumerate(parts):
                    if r != self.root and e > b:
                        ops.append(dist.P2POp(dist.isend, f[b:e], self._peer(r), self.group))
                mine.append(f[lo:hi])
            return (dist.batch_isend_irecv(ops) if ops else []), mine
        mine = [torch.empty((hi - lo, self.width), dtype=torch.int32, device=self.device) for _ in fulls]
        ops = [dist.P2POp(dist.irecv, m, self._peer(self.root), self.group) for m in mine] if hi > lo else []
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        return [], mine

    def _post_gather(self, parts):
        """Root only: the full output tensor and the (un-waited) receives of every other rank's rows into views of it."""
        full = torch.empty((parts[-1][1], self.width), dtype=torch.int32, device=self.device)
        ops = [dist.P2POp(dist.irecv, full[b:e], self._peer(r), self.group)
               for r, (b, e) in enumerate(parts) if r != self.root and e > b]
        return full, (dist.batch_isend_irecv(ops) if ops else [])

    def run(self, op, in0, in1, count, sync=None):
        """in0/in1: full [count, n+1] int32 tensors on the root (ignored elsewhere; in1 None for a unary gate).
        Returns the full output on the root, None on the other ranks.  With `sync` (a callable that drains this rank's
        device queue) the scatter / compute / gather times of this call are left in self.last_timing (seconds).

        The root does not serialise scatter -> compute -> gather (round 4): it POSTS the sends, launches the bootstrap of its own shard
        (asynchronous on its stream), posts the receives of the results and only then waits -- its own shard never waits for the other
        ranks' rows to leave.  The receives are posted AFTER the root's own launch on purpose: RCCL's receive kernel spins on its CUs until
        the peers' data arrives (a whole batch later), and the bootstrap kernel needs every CU of the chip to itself (one workgroup per CU,
        all of its LDS and registers) -- posted first, the receive would hold CUs that the root's own workgroups then queue behind.
        On the root "scatter_s" is therefore the time to post, and "gather_s" the wait that remains after its own compute."""
        import time
        count, has_in1 = self._header(count, in1 is not None)
        parts = partition(count, self.world)
        lo, hi = parts[self.rank]
        root = self.rank == self.root
        t0 = time.perf_counter()
        sends, recvs, full = [], [], None
        if self.world == 1:
            mine = [t.to(self.device) for t in ([in0] + ([in1] if has_in1 else []))]
        else:
            fulls = ([in0] + ([in1] if has_in1 else [])) if root else [None] * (2 if has_in1 else 1)
            sends, mine = self._post_scatter(fulls, parts)
        if sync and not root:
            sync()
        t1 = time.perf_counter()
        if hi > lo:
            out = self.compute(op, mine[0].contiguous(), mine[1].contiguous() if has_in1 else None)
        else:
            out = torch.empty((0, self.width), dtype=torch.int32, device=self.device)
        if sync:
            sync()
        t2 = time.perf_counter()
        if self.world == 1:
            full = out
        elif root:
            full, recvs = self._post_gather(parts)
            full[lo:hi] = out
            for w in sends + recvs:
                w.wait()
        else:
            if hi > lo:
                for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, out, self._peer(self.root), self.group)]):
                    w.wait()
            full = None
        if sync:
            sync()
            self.last_timing = {"scatter_s": t1 - t0, "compute_s": t2 - t1, "gather_s": time.perf_counter() - t2}
        return full


def engine_compute(engine, gpu=None):
    """compute callback running the HIP path on this rank's GPU.  Tensors arrive on the communication device: device
    tensors are used in place (nccl); CPU tensors (gloo rehearsal) are staged to `gpu` and back."""
    def fn(op, a, b):
        nonlocal gpu
        if gpu is None:
            gpu = torch.device("cuda", engine.device)
        st = torch.cuda.current_stream().cuda_stream
        if a.is_cuda:
            out = torch.empty_like(a)
            engine.gate_batch_dev(op, a, b, out, a.shape[0], st)
            return out
        da, db = a.to(gpu), (None if b is None else b.to(gpu))
        out = torch.empty_like(da)
        engine.gate_batch_dev(op, da, db, out, da.shape[0], st)
        return out.cpu()
    return fn
