"""The multi-GPU exchange through the library's own RCCL entry points (include/tracs_hip.h part 4, csrc/comm.cpp).

`RcclDist` offers the handful of `torch.distributed` calls tracs_amd/partition.py and multigpu.py make -- all_gather of row
panels, all_reduce, broadcast, send / recv, barrier, broadcast_object_list -- over a `tracs_comm`: the data path of the
multi-GPU runs is libtracs_hip.so + RCCL, and torch only launches the processes and lends the rendezvous store that carries the
128-byte communicator id (and small Python objects) between them.  The collectives run on a stream of their own, ordered
against the caller's stream with events, so an exchange enqueued with async_op=True overlaps the kernels that follow it.

The CPU tests (gloo, several ranks on one host) keep using torch.distributed itself: RCCL wants one GPU per rank.
"""
import ctypes as C
import os
import pickle

from . import _lib


class _Work:
    def __init__(self, event):
        self._event = event

    def wait(self):
        import torch
        torch.cuda.current_stream().wait_event(self._event)
        return True


class _ReduceOp:
    SUM, MAX, MIN = 0, 1, 2


def _env_store():
    """The rendezvous store of the launcher (torchrun's agent store, or a TCPStore on MASTER_ADDR:MASTER_PORT hosted by rank 0)."""
    import torch.distributed as dist
    store, rank, world = next(dist.rendezvous("env://"))
    return store, rank, world


class RcclDist:
    """A tracs_comm behind the subset of the torch.distributed interface the multi-GPU drivers use."""
    ReduceOp = _ReduceOp

    def __init__(self, device, store=None, rank=None, world=None, tag="tracs"):
        import torch
        self._L = _lib.require_gpu()
        if store is None:
            store, rank, world = _env_store()
        self.store, self.rank, self.world, self.device = store, int(rank), int(world), device
        torch.cuda.set_device(device)
        key = "%s/comm_id" % tag
        if self.rank == 0:
            buf = C.create_string_buffer(128)
            _lib.check(self._L.tracs_comm_unique_id(buf, 128))
            store.set(key, bytes(buf.raw))
        uid = bytes(store.get(key))
        self._h = C.c_void_p()
        _lib.check(self._L.tracs_comm_create(uid, self.rank, self.world, C.byref(self._h)))
        self.stream = torch.cuda.Stream(device=device)
        self._seq = 0
        self._tag = tag

    # ---- ordering against the caller's stream ---------------------------------------------------------------------------------
    def _enter(self):
        import torch
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self.stream.wait_event(ev)
        return C.c_void_p(self.stream.cuda_stream)

    def _leave(self, async_op):
        import torch
        ev = torch.cuda.Event()
        ev.record(self.stream)
        w = _Work(ev)
        if async_op:
            return w
        w.wait()
        return None

    # ---- collectives --------------------------------------------------------------------------------------------------------------
    def all_gather(self, outs, inp, async_op=False):
        """outs[q]: where rank q's block lands (equal sizes, views of ONE allocation laid out alike on every rank, as the row
        panels of a pair matrix are); inp: this rank's block (copied into outs[rank] first unless it already is that memory)."""
        assert len(outs) == self.world
        nbytes = inp.numel() * inp.element_size()
        for o in outs:
            assert o.is_contiguous() and o.numel() * o.element_size() == nbytes
        if outs[self.rank].data_ptr() != inp.data_ptr():
            outs[self.rank].copy_(inp)
        base = min(o.data_ptr() for o in outs)
        offs = (C.c_size_t * self.world)(*[o.data_ptr() - base for o in outs])
        s = self._enter()
        _lib.check(self._L.tracs_allgather_panels(self._h, C.c_void_p(base), offs, nbytes, s))
        return self._leave(async_op)

    _DTYPES = {"torch.int64": 0, "torch.float64": 1, "torch.int32": 2, "torch.uint8": 3}

    def all_reduce(self, t, op=_ReduceOp.SUM, async_op=False):
        import torch
        assert t.is_contiguous()
        code = self._DTYPES.get(str(t.dtype))
        if str(t.dtype) == "torch.int32":
            # (uint32 on the wire: sums and maxima of non-negative int32 values agree)
            assert op == _ReduceOp.SUM or bool((t >= 0).all())
        if code is None:
            raise TypeError("RcclDist.all_reduce: int64, float64, int32 (non-negative) or uint8 tensors")
        s = self._enter()
        _lib.check(self._L.tracs_allreduce(self._h, C.c_void_p(t.data_ptr()), t.numel(), code, int(op), s))
        return self._leave(async_op)

    def reduce_scatter_rows(self, m, rows_per_rank, op=_ReduceOp.SUM, async_op=False):
        """m: [world * rows_per_rank, ld] (contiguous); afterwards rows [rank * rows_per_rank, (rank + 1) * rows_per_rank) of this
        rank's m hold the reduction over the ranks of those rows (tracs_reduce_scatter, in place)."""
        assert m.is_contiguous() and m.shape[0] == self.world * rows_per_rank
        code = self._DTYPES.get(str(m.dtype))
        if code is None:
            raise TypeError("RcclDist.reduce_scatter_rows: int64, float64, int32 or uint8")
        s = self._enter()
        _lib.check(self._L.tracs_reduce_scatter(self._h, C.c_void_p(m.data_ptr()), rows_per_rank * m.shape[1], code, int(op), s))
        return self._leave(async_op)

    def all_to_all_blocks(self, send, recv, block_bytes, async_op=False):
        """send / recv: contiguous uint8 device tensors of world * block_bytes; block q of send -> rank q, where it becomes block
        `rank` of recv (tracs_alltoall: every pair of ranks over its own xGMI link)."""
        assert send.is_contiguous() and recv.is_contiguous()
        assert send.numel() * send.element_size() == self.world * block_bytes == recv.numel() * recv.element_size()
        s = self._enter()
        _lib.check(self._L.tracs_alltoall(self._h, C.c_void_p(send.data_ptr()), C.c_void_p(recv.data_ptr()), int(block_bytes), s))
        return self._leave(async_op)

    def ranks_seen(self):
        """(rank, world) as RCCL itself reports them for this communicator (ncclCommUserRank / ncclCommCount)."""
        return int(self._L.tracs_comm_rank(self._h)), int(self._L.tracs_comm_world(self._h))

    def rccl_version(self):
        return int(self._L.tracs_rccl_version())

    def get_backend(self):
        return "rccl"

    def broadcast(self, t, src=0, async_op=False):
        assert t.is_contiguous()
        s = self._enter()
        _lib.check(self._L.tracs_bcast(self._h, C.c_void_p(t.data_ptr()), t.numel() * t.element_size(), int(src), s))
        return self._leave(async_op)

    def broadcast_planes(self, aln, src=0):
        """the packed planes of rank `src` -> every rank's handle (tracs_bcast_planes: marks the receiving handles packed)"""
        import torch
        s = self._enter()
        _lib.check(self._L.tracs_bcast_planes(self._h, aln._h, int(src), s))
        self._leave(False)
        torch.cuda.synchronize()

    def send(self, t, dst):
        assert t.is_contiguous()
        s = self._enter()
        _lib.check(self._L.tracs_send(self._h, C.c_void_p(t.data_ptr()), t.numel() * t.element_size(), int(dst), s))
        self._leave(False)

    def recv(self, t, src):
        assert t.is_contiguous()
        s = self._enter()
        _lib.check(self._L.tracs_recv(self._h, C.c_void_p(t.data_ptr()), t.numel() * t.element_size(), int(src), s))
        self._leave(False)

    def barrier(self):
        import torch
        t = torch.ones(1, dtype=torch.int64, device=self.device)
        self.all_reduce(t)
        torch.cuda.synchronize()
        assert int(t.item()) == self.world

    # ---- small Python objects: through the store ------------------------------------------------------------------------------------
    def broadcast_object_list(self, objs, src=0):
        self._seq += 1
        key = "%s/obj/%d" % (self._tag, self._seq)
        if self.rank == src:
            self.store.set(key, pickle.dumps(list(objs)))
        got = pickle.loads(bytes(self.store.get(key)))
        for k in range(len(objs)):
            objs[k] = got[k]

    def self_test(self):
        """One all-reduce, one in-place all-gather, one broadcast, one in-place reduce-scatter (with a uint32 wrap) and one all-to-all
        on tiny buffers, checked, and the communicator's own idea of its size: a broken exchange is found before any result depends
        on it."""
        import torch
        if self.ranks_seen() != (self.rank, self.world):
            return False
        W = self.world
        rs = torch.full((W * 4, 8), self.rank + 1, dtype=torch.int32, device=self.device)
        self.reduce_scatter_rows(rs, 4)
        wrap = torch.full((W * 2, 4), -1 if self.rank == 0 else 3, dtype=torch.int32, device=self.device)     # 0xFFFFFFFF + 3 (W - 1) wraps
        self.reduce_scatter_rows(wrap, 2)
        a2a_s = (torch.arange(W, device=self.device, dtype=torch.int32)[:, None] * 100 + self.rank).repeat(1, 16).contiguous()
        a2a_r = torch.zeros_like(a2a_s)
        self.all_to_all_blocks(a2a_s.view(torch.uint8).view(-1), a2a_r.view(torch.uint8).view(-1), 64)
        torch.cuda.synchronize()
        if not bool((rs[self.rank * 4:(self.rank + 1) * 4] == W * (W + 1) // 2).all()):
            return False
        want = (0xFFFFFFFF + 3 * (W - 1)) & 0xFFFFFFFF
        if not bool(((wrap[self.rank * 2:(self.rank + 1) * 2].to(torch.int64) & 0xFFFFFFFF) == want).all()):
            return False
        if not bool((a2a_r == (torch.arange(W, device=self.device, dtype=torch.int32)[:, None] + self.rank * 100)).all()):
            return False
        v = torch.tensor([self.rank + 1], dtype=torch.int64, device=self.device)
        self.all_reduce(v)
        ok = int(v.item()) == self.world * (self.world + 1) // 2
        buf = torch.zeros((self.world, 64), dtype=torch.int32, device=self.device)
        buf[self.rank] = self.rank + 7
        self.all_gather([buf[q] for q in range(self.world)], buf[self.rank])
        torch.cuda.synchronize()
        ok = ok and bool((buf == (torch.arange(self.world, device=self.device, dtype=torch.int32) + 7)[:, None]).all())
        b = torch.full((16,), 3 if self.rank == 0 else 0, dtype=torch.float64, device=self.device)
        self.broadcast(b, src=0)
        torch.cuda.synchronize()
        return ok and bool((b == 3).all())

    def destroy_process_group(self):
        self.close()

    def close(self):
        if getattr(self, "_h", None):
            self._L.tracs_comm_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def backend_choice(world, ndev, env_name):
    """'rccl' (the library's own RCCL entry points), 'nccl' (torch.distributed over RCCL) or 'gloo': the environment variable
    `env_name` when set, else rccl when every rank has a GPU of its own, else gloo (several ranks on one GPU: tests)."""
    v = os.environ.get(env_name)
    if v:
        return v
    return "rccl" if ndev >= world else "gloo"
