"""TEST ONLY: the slab borders over torch.distributed (gloo, host buffers) -- the transport of the world_size-2 CPU tests the build contract
asks for (tests/test_slab_hostsim.py).  The package itself has no PyTorch in it (medpy_amd/slab.py: RcclExchange, StoreExchange over
medpy_amd.rendezvous.FileStore)."""
import numpy as np


class DistExchange(object):
    """One slab per process; neighbours are rank-1 / rank+1 of ``torch.distributed``; the packed borders
    travel as torch tensors.  Used with "gloo" (host buffers): the CPU test tier over the host simulator,
    and a development mode of bench.py.  The production transport is RcclExchange below -- PyTorch's ROCm
    wheel bundles its own HIP runtime, which cannot share a process with the system runtime this library
    is linked against, so torch.cuda tensors are deliberately not used here."""

    def __init__(self, slab, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.slabs = [slab]
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.group = group
        self.on_device = dist.get_backend(group) == "nccl"
        self.dev = torch.device("cuda", torch.cuda.current_device()) if self.on_device else torch.device("cpu")
        self._bufs = {}

    def _buf(self, key, nbytes):
        b = self._bufs.get(key)
        if b is None or b.numel() != nbytes:
            b = self.torch.zeros(nbytes, dtype=self.torch.uint8, device=self.dev)
            self._bufs[key] = b
        return b

    def _raw(self, t):
        return t.data_ptr() if self.on_device else t.numpy()

    # -- what medpy_amd.slab.HostTransport hands to the schedule (mgc_solve over the slab group) as the callbacks of an mgc_transport
    def xchg(self, lo, hi):
        torch, dist = self.torch, self.dist
        ops, out = [], [None, None]
        for k, (peer, data) in enumerate(((self.rank - 1, lo), (self.rank + 1, hi))):
            if data is None:
                continue
            snd = torch.frombuffer(bytearray(data), dtype=torch.uint8)
            out[k] = torch.zeros(len(data), dtype=torch.uint8)
            ops.append(dist.P2POp(dist.isend, snd, peer, self.group))
            ops.append(dist.P2POp(dist.irecv, out[k], peer, self.group))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        return tuple(None if t is None else t.numpy().tobytes() for t in out)

    def allreduce_i64(self, a, op):
        t = self.torch.as_tensor(np.asarray(a, dtype=np.int64))
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN if op == 1 else self.dist.ReduceOp.SUM, group=self.group)
        return t.numpy()

    def send(self, side, data):
        self.dist.send(self.torch.frombuffer(bytearray(data), dtype=self.torch.uint8), self.rank + (1 if side else -1), group=self.group)

    def recv(self, side, nbytes):
        t = self.torch.zeros(int(nbytes), dtype=self.torch.uint8)
        self.dist.recv(t, self.rank + (1 if side else -1), group=self.group)
        return t.numpy().tobytes()

    def allreduce_sum(self, values):
        t = self.torch.as_tensor(np.sum(np.asarray(values, dtype=np.float64), axis=0), dtype=self.torch.float64).reshape(-1).to(self.dev)
        self.dist.all_reduce(t, group=self.group)
        out = t.cpu().numpy()
        return out if out.size > 1 else float(out[0])

    def allreduce_max(self, values):
        t = self.torch.as_tensor(np.max(np.asarray(values, dtype=np.float64), axis=0), dtype=self.torch.float64).reshape(-1).to(self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        return t.cpu().numpy()
