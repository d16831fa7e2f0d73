"""Native data-parallel collective: the `spn_comm_*` entry points of libspn.so (RCCL bound at run time, see csrc/comm.cpp).

`NativeComm` is the communicator `parallel.GradSync(transport="spn")` uses instead of `torch.distributed.all_reduce`: one
`ncclAllReduce` per gradient bucket on a dedicated communication stream, fenced with HIP events against the stream that runs the
backward.  The 128-byte RCCL id is made on rank 0 and handed to the other ranks through whatever channel the launcher already has
(`torch.distributed.broadcast_object_list` over the job's process group, or a file)."""
from __future__ import annotations

import ctypes
import os
from typing import Optional

import torch

from .lib import SpnError, call, load, ptr, stream_ptr, c_int


def rccl_path() -> Optional[bytes]:
    """The RCCL copy PyTorch ships (so that libspn.so binds to the runtime that is already in the process)."""
    cand = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    return cand.encode() if os.path.exists(cand) else None


def available() -> None:
    """Raises SpnError when this process cannot bind RCCL through libspn.so; no collective, no id, no bootstrap thread."""
    call("spn_comm_available", rccl_path())


def unique_id() -> bytes:
    buf = ctypes.create_string_buffer(128)
    call("spn_comm_unique_id", buf, rccl_path())
    return buf.raw


class NativeComm:
    def __init__(self, world: int, rank: int, uid: bytes):
        assert len(uid) == 128
        load()
        self.world, self.rank = world, rank
        self._h = ctypes.c_void_p()
        call("spn_comm_init", ctypes.byref(self._h), c_int(world), c_int(rank), ctypes.c_char_p(uid), rccl_path())

    @classmethod
    def from_group(cls, group=None) -> "NativeComm":
        """Collective over an initialised torch.distributed group: rank 0 makes the id, everybody receives it."""
        import torch.distributed as dist
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        box, err = [None], None
        if rank == 0:
            try:
                box = [unique_id()]
            except Exception as exc:  # noqa: BLE001 -- the broadcast below must still happen: the other ranks are waiting in it
                err = exc
        if world > 1:
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        if box[0] is None:
            raise SpnError(f"rank 0 could not make an RCCL id: {err!r}" if rank == 0 else "rank 0 could not make an RCCL id")
        return cls(world, rank, box[0])

    def all_reduce_(self, t: torch.Tensor) -> None:
        """In-place sum over the ranks, asynchronous: ordered behind the current stream, runs on the communication stream."""
        assert t.is_cuda and t.is_contiguous() and t.dtype in (torch.float32, torch.bfloat16)
        call("spn_comm_allreduce", self._h, ptr(t), ctypes.c_size_t(t.numel()), c_int(0 if t.dtype == torch.float32 else 1), stream_ptr())

    def wait(self) -> None:
        """The current stream waits (on the device) for every all-reduce enqueued so far."""
        call("spn_comm_wait", self._h, stream_ptr())

    def close(self) -> None:
        if self._h:
            call("spn_comm_destroy", self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        # at interpreter exit the library (or ctypes itself) may already be torn down: never raise from here, and skip the call
        # when the binding is gone rather than jump into an unloaded library
        try:
            from . import lib as _lib
            if _lib is not None and getattr(_lib, "_lib", None) is not None and self._h:
                self.close()
        except BaseException:   # noqa: BLE001
            pass
