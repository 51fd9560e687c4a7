"""Sharding of a gate batch over the GPUs of one node.

Gates are independent: rank g bootstraps the contiguous range [g*B/G, (g+1)*B/G) with its own replica of the
keys.  The only communication is the scatter of input ciphertexts from the root and the gather of results back
(RCCL over xGMI when the process group's backend is "nccl"; the same code runs on "gloo" for CPU tests).
There is no reduction and no exchange inside the path.
"""
import torch
import torch.distributed as dist


def partition(count, world):
    """Contiguous ranges, sizes differ by at most one."""
    return [(count * g // world, count * (g + 1) // world) for g in range(world)]


class ShardedGates:
    """compute(op, in0, in1) -> out works on this rank's shard (tensors int32 [k, n+1] on `device`)."""

    def __init__(self, compute, width, device, group=None, root=0):
        self.compute, self.width, self.device, self.group, self.root = compute, width, device, group, root
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1

    def _scatter(self, full, count):
        parts = partition(count, self.world)
        cap = max(e - b for b, e in parts)
        mine = torch.zeros((cap, self.width), dtype=torch.int32, device=self.device)
        if self.world == 1:
            mine.copy_(full.to(self.device))
            return mine, parts
        chunks = None
        if self.rank == self.root:
            chunks = []
            for b, e in parts:
                c = torch.zeros((cap, self.width), dtype=torch.int32, device=self.device)
                c[: e - b] = full[b:e].to(self.device)
                chunks.append(c)
        dist.scatter(mine, chunks, src=self.root, group=self.group)
        return mine, parts

    def run(self, op, in0, in1, count):
        """in0/in1: full [count, n+1] int32 tensors on the root (ignored elsewhere).  Returns the full output on
        the root, None on the other ranks."""
        cnt = torch.tensor([count if self.rank == self.root else 0], dtype=torch.int64, device=self.device)
        if self.world > 1:
            dist.broadcast(cnt, src=self.root, group=self.group)
        count = int(cnt.item())
        a, parts = self._scatter(in0, count)
        b, _ = self._scatter(in1, count) if in1 is not None or self.rank != self.root else (None, None)
        lo, hi = parts[self.rank]
        k = hi - lo
        out = torch.zeros_like(a)
        if k:
            out[:k] = self.compute(op, a[:k].contiguous(), None if b is None else b[:k].contiguous())
        if self.world == 1:
            return out[:k]
        gathered = [torch.zeros_like(out) for _ in range(self.world)] if self.rank == self.root else None
        dist.gather(out, gathered, dst=self.root, group=self.group)
        if self.rank != self.root:
            return None
        return torch.cat([g[: e - b] for g, (b, e) in zip(gathered, parts)], dim=0)


def engine_compute(engine):
    """compute callback running the HIP path on this rank's GPU (device tensors in, device tensor out)."""
    def fn(op, a, b):
        out = torch.empty_like(a)
        st = torch.cuda.current_stream().cuda_stream
        engine.gate_batch_dev(op, a, b, out, a.shape[0], st)
        return out
    return fn
