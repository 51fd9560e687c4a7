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

    def _scatter(self, fulls, parts):
        """fulls: list of full [count, width] tensors on the root (None elsewhere).  Returns this rank's rows of each."""
        lo, hi = parts[self.rank]
        if self.rank == self.root:
            ops, mine = [], []
            for f in fulls:
                assert f.dtype == torch.int32 and f.is_contiguous() and f.device == torch.device(self.device)
                for r, (b, e) in enumerate(parts):
                    if r != self.root and e > b:
                        ops.append(dist.P2POp(dist.isend, f[b:e], self._peer(r), self.group))
                mine.append(f[lo:hi])
        else:
            mine = [torch.empty((hi - lo, self.width), dtype=torch.int32, device=self.device) for _ in fulls]
            ops = [dist.P2POp(dist.irecv, m, self._peer(self.root), self.group) for m in mine] if hi > lo else []
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        return mine

    def _gather(self, out, parts):
        """out: this rank's [k, width] result.  Returns the full tensor on the root, None elsewhere."""
        lo, hi = parts[self.rank]
        if self.rank != self.root:
            if hi > lo:
                for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, out, self._peer(self.root), self.group)]):
                    w.wait()
            return None
        full = torch.empty((parts[-1][1], self.width), dtype=torch.int32, device=self.device)
        full[lo:hi] = out
        ops = [dist.P2POp(dist.irecv, full[b:e], self._peer(r), self.group)
               for r, (b, e) in enumerate(parts) if r != self.root and e > b]
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        return full

    def run(self, op, in0, in1, count, sync=None):
        """in0/in1: full [count, n+1] int32 tensors on the root (ignored elsewhere; in1 None for a unary gate).
        Returns the full output on the root, None on the other ranks.  With `sync` (a callable that drains this rank's
        device queue) the scatter / compute / gather times of this call are left in self.last_timing (seconds)."""
        import time
        count, has_in1 = self._header(count, in1 is not None)
        parts = partition(count, self.world)
        lo, hi = parts[self.rank]
        t0 = time.perf_counter()
        if self.world == 1:
            mine = [t.to(self.device) for t in ([in0] + ([in1] if has_in1 else []))]
        else:
            fulls = ([in0] + ([in1] if has_in1 else [])) if self.rank == self.root else [None] * (2 if has_in1 else 1)
            mine = self._scatter(fulls, parts)
        if sync:
            sync()
        t1 = time.perf_counter()
        if hi > lo:
            out = self.compute(op, mine[0].contiguous(), mine[1].contiguous() if has_in1 else None)
        else:
            out = torch.empty((0, self.width), dtype=torch.int32, device=self.device)
        if sync:
            sync()
        t2 = time.perf_counter()
        full = out if self.world == 1 else self._gather(out, parts)
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
