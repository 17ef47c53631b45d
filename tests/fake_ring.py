"""Test infrastructure: an in-process stand-in for an RCCL ring.

RCCL refuses two ranks on one device and no multi-GPU node exists for the builder, so the native slab loops
(``percnn_pi_slab_rollout_fwd/bwd_*``) had only ever driven their ``percnn_pi_halo_ring`` callbacks -- ``ncclGroupStart``,
``ncclSend``, ``ncclRecv``, ``ncclGroupEnd`` -- with a rank that talks to itself.  Here every "rank" is a THREAD of this process with
its own HIP stream and its own local slab; the four function pointers of its ring are ctypes callbacks with RCCL's semantics:

* ``send(buf, count, dtype, peer, comm, stream)``: the bytes are copied (on ``stream``) into a staging buffer, an event is
  recorded behind the copy, and (buffer, event) goes into the FIFO of the ordered pair (me -> peer) -- per peer, sends and
  receives pair up in issue order, exactly the rule that matters when prev == next (two ranks);
* ``recv(buf, count, dtype, peer, comm, stream)``: takes the oldest entry of (peer -> me), makes ``stream`` wait for its event
  and copies it into ``buf``; count / dtype mismatches between the paired operations are errors (ncclInvalidArgument);
* operations are only legal between ``group_start`` and ``group_end`` (what RCCL requires of send / recv pairs issued by one
  thread to more than one peer), and a group must be closed before the next one opens;
* a callback may be told to FAIL (``fail_at``): the native loop must hand that code back to its caller.

Everything is ordered by streams and events only -- no host synchronisation inside an exchange."""
import ctypes
import queue
import threading

import torch

from percnn_amd import _lib
from percnn_amd.slab import HaloExchanger

_hip = None


def hip():
    global _hip
    if _hip is None:
        _hip = ctypes.CDLL("libamdhip64.so")
        _hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
    return _hip


NCCL_F32, NCCL_F64 = 7, 8
ESZ = {NCCL_F32: 4, NCCL_F64: 8}


class FakeFabric:
    """what the ranks of one ring share: the FIFOs of the ordered pairs, the reduction of the gradient blocks, the log"""

    def __init__(self, world: int, take_timeout_s: float = 20.0):
        self.world = world
        self.fifo = {(a, b): queue.Queue() for a in range(world) for b in range(world)}
        self.take_timeout_s = take_timeout_s
        self.barrier = threading.Barrier(world)
        self.lock = threading.Lock()
        self.red = None
        self.log = []                           # (rank, op, peer, count) in issue order per rank
        self.errors = []

    def all_reduce_sum(self, rank: int, t: torch.Tensor) -> torch.Tensor:
        torch.cuda.current_stream().synchronize()
        with self.lock:
            self.red = t.clone() if self.red is None else self.red + t
        self.barrier.wait()
        out = self.red.clone()
        self.barrier.wait()
        if rank == 0:
            self.red = None
        self.barrier.wait()
        t.copy_(out)
        return t


class FakeRingExchanger(HaloExchanger):
    """drives the native slab loops through a percnn_pi_halo_ring whose callbacks are Python"""

    GROUP_START = ctypes.CFUNCTYPE(ctypes.c_int)
    SEND = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p)

    def __init__(self, fabric: FakeFabric, rank: int, fail_at=None, packed: bool = True):
        self.group, self.force_p2p = None, True
        self.fabric, self.rank, self.world = fabric, rank, fabric.world
        self.prev, self.next = (rank - 1) % self.world, (rank + 1) % self.world
        self._bufs = {}
        self.in_group = False
        self.groups = 0
        self.ops = 0
        self.fail_at = fail_at                  # (op name, n): the n-th call of that callback returns ncclSystemError (2)
        self.calls = {"group_start": 0, "send": 0, "recv": 0, "group_end": 0}
        self.packed = packed
        self._stage = None
        self._cb = (self.GROUP_START(self._group_start), self.GROUP_START(self._group_end), self.SEND(self._send), self.SEND(self._recv))
        self._ring = None

    # ---- the four "RCCL" entry points ---------------------------------------------------------------------------------
    def _fails(self, name):
        self.calls[name] += 1
        return self.fail_at is not None and self.fail_at[0] == name and self.calls[name] == self.fail_at[1]

    def _group_start(self):
        if self._fails("group_start"):
            return 2
        if self.in_group:
            self.fabric.errors.append((self.rank, "nested group"))
            return 4
        self.in_group = True
        return 0

    def _group_end(self):
        if self._fails("group_end"):
            return 2
        if not self.in_group:
            self.fabric.errors.append((self.rank, "group_end without group_start"))
            return 4
        self.in_group = False
        self.groups += 1
        return 0

    def _send(self, buf, count, dtype, peer, comm, stream):
        try:
            if self._fails("send"):
                return 2
            if not self.in_group or comm != 0xC0FFEE or dtype not in ESZ or not (0 <= peer < self.world):
                self.fabric.errors.append((self.rank, "send outside a group / bad comm / dtype / peer", peer))
                return 4
            nbytes = count * ESZ[dtype]
            stage = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
            st = torch.cuda.ExternalStream(stream) if stream else torch.cuda.default_stream()
            with torch.cuda.stream(st):
                stage.record_stream(st)
                if hip().hipMemcpyAsync(stage.data_ptr(), buf, nbytes, 3, stream) != 0:
                    return 1
                ev = torch.cuda.Event()
                ev.record(st)
            self.fabric.log.append((self.rank, "send", peer, count))
            self.fabric.fifo[(self.rank, peer)].put((stage, ev, count, dtype))
            self.ops += 1
            return 0
        except Exception as e:                  # never let an exception cross the C frame
            self.fabric.errors.append((self.rank, "send raised", repr(e)))
            return 3

    def _recv(self, buf, count, dtype, peer, comm, stream):
        try:
            if self._fails("recv"):
                return 2
            if not self.in_group or comm != 0xC0FFEE or dtype not in ESZ or not (0 <= peer < self.world):
                self.fabric.errors.append((self.rank, "recv outside a group / bad comm / dtype / peer", peer))
                return 4
            try:
                stage, ev, scount, sdtype = self.fabric.fifo[(peer, self.rank)].get(timeout=self.fabric.take_timeout_s)
            except queue.Empty:
                self.fabric.errors.append((self.rank, "recv: no matching send", peer))
                return 6
            if scount != count or sdtype != dtype:
                self.fabric.errors.append((self.rank, "recv: count / dtype differ from the paired send", peer, count, scount))
                return 4
            st = torch.cuda.ExternalStream(stream) if stream else torch.cuda.default_stream()
            with torch.cuda.stream(st):
                st.wait_event(ev)
                if hip().hipMemcpyAsync(buf, stage.data_ptr(), count * ESZ[dtype], 3, stream) != 0:
                    return 1
                stage.record_stream(st)
            self.fabric.log.append((self.rank, "recv", peer, count))
            self.ops += 1
            return 0
        except Exception as e:
            self.fabric.errors.append((self.rank, "recv raised", repr(e)))
            return 3

    # ---- HaloExchanger interface ----------------------------------------------------------------------------------------
    @property
    def ranks_seen(self):
        return self.world

    def native_ring(self):
        if self._ring is None:
            addr = lambda fn: ctypes.cast(fn, ctypes.c_void_p).value
            self._ring = _lib.HaloRing(0xC0FFEE, self.prev, self.next, NCCL_F32, NCCL_F64, addr(self._cb[0]), addr(self._cb[1]),
                                       addr(self._cb[2]), addr(self._cb[3]), None, None, 0)
        st = self._stage
        self._ring.stage = st.data_ptr() if st is not None else None
        self._ring.stage_bytes = st.numel() if st is not None else 0
        return True, ctypes.byref(self._ring)

    def prepare(self, slab, halo):
        if not self.packed:
            self._stage = None
            return
        need = 8 * halo * slab[0, 0].numel() * slab.element_size()
        if self._stage is None or self._stage.numel() < need:
            self._stage = torch.empty(need, dtype=torch.uint8, device=slab.device)

    def all_reduce_sum_(self, t):
        return self.fabric.all_reduce_sum(self.rank, t)
