"""ThreadComm -- several ranks inside ONE process, one thread each, with torch.distributed's call surface for the three
collectives the sharded path uses.  TEST INFRASTRUCTURE: a GPU box admits at most six processes on its card, so an
8-rank geometry (BASELINE config C4: 2048 Doppler bins, 256 per rank) is rehearsed on one device as eight threads, each
with its own library handle and stream.  The collectives block the calling thread (the stream the tensors were produced
on is synchronised first), which is the semantics of a collective followed by a synchronisation."""
import threading
import types

import torch


class ThreadWorld:
    def __init__(self, world):
        self.world = world
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world


class ThreadComm:
    ReduceOp = types.SimpleNamespace(SUM='sum', MAX='max')

    def __init__(self, world, rank):
        self.w, self.rank = world, rank

    # -- the queries -----------------------------------------------------------------------------
    def get_rank(self, group=None):
        return self.rank

    def get_world_size(self, group=None):
        return self.w.world

    def get_backend(self, group=None):
        return 'threads'

    def new_group(self, backend=None):
        return None

    # -- collectives -----------------------------------------------------------------------------
    @staticmethod
    def _sync(t):
        if t.is_cuda:
            torch.cuda.current_stream(t.device).synchronize()

    def barrier(self, group=None):
        self.w.barrier.wait()

    def broadcast(self, tensor, src=0, group=None):
        self._sync(tensor)
        if self.rank == src:
            self.w.slots[src] = tensor
        self.w.barrier.wait()
        if self.rank != src:
            tensor.copy_(self.w.slots[src])
            self._sync(tensor)
        self.w.barrier.wait()

    def all_gather_into_tensor(self, out, inp, group=None):
        self._sync(inp)
        self.w.slots[self.rank] = inp
        self.w.barrier.wait()
        n = inp.shape[0]
        for r in range(self.w.world):
            out[r * n:(r + 1) * n].copy_(self.w.slots[r])
        self._sync(out)
        self.w.barrier.wait()

    def all_reduce(self, tensor, op='sum', group=None):
        self._sync(tensor)
        self.w.slots[self.rank] = tensor.clone()
        self._sync(tensor)
        self.w.barrier.wait()
        acc = self.w.slots[0].clone()
        for r in range(1, self.w.world):          # fixed order: the same bits on every rank
            acc = torch.maximum(acc, self.w.slots[r]) if op == 'max' else acc + self.w.slots[r]
        tensor.copy_(acc)
        self._sync(tensor)
        self.w.barrier.wait()


def run_ranks(world, body):
    """Run ``body(comm)`` on ``world`` threads; re-raises the first failure.  Returns the list of results by rank."""
    w = ThreadWorld(world)
    out, errs = [None] * world, []

    def target(r):
        try:
            out[r] = body(ThreadComm(w, r))
        except BaseException as e:      # noqa: BLE001
            errs.append(e)
            w.barrier.abort()
    ts = [threading.Thread(target=target, args=(r,), name=f'rank{r}') for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    if errs:
        real = [e for e in errs if not isinstance(e, threading.BrokenBarrierError)]
        raise (real or errs)[0]
    return out
