"""The library-owned RCCL communicator (``mh_comm_*``, csrc/comm.cpp) behind the gradient exchange of the data-parallel step
(reference: DDP under Lightning, train.py:461-474).

``torch.distributed`` stays the rendezvous: rank 0 creates the 128-byte RCCL id and the process group that ``bench.py`` /
the launcher already set up carries it to the other ranks (one ``broadcast_object_list`` on whatever backend that group has);
from then on the collectives are issued by ``libmidihip.so`` on the caller's HIP streams -- the same call sequence a host in
another language would make through the C-ABI.  Select it with ``TrainMIDIModel.use_comm(MHComm.from_process_group(...))`` or
``bench.py --comm mh``; the default exchange remains ``torch.distributed`` ("nccl" = RCCL)."""
from __future__ import annotations

import ctypes
from typing import Optional

import sys

import torch

from .lib import MH_BF16, MH_F32, lib


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.bfloat16:
        return MH_BF16
    if t.dtype == torch.float32:
        return MH_F32
    raise TypeError(f"MHComm: {t.dtype} is not exchanged (bf16 / fp32 gradients and parameters only)")


class MHComm:
    """One RCCL communicator of this process, created by ``mh_comm_init``."""

    def __init__(self, rank: int, world: int, device: int, unique_id: Optional[bytes] = None):
        if unique_id is None:
            if world != 1:
                raise ValueError("MHComm: every rank needs rank 0's unique id (MHComm.from_process_group ships it)")
            unique_id = self.new_unique_id()
        if len(unique_id) != 128:
            raise ValueError("MHComm: the RCCL unique id is 128 bytes")
        self.rank, self.world, self.device = int(rank), int(world), int(device)
        self._h = ctypes.c_void_p()
        buf = ctypes.create_string_buffer(bytes(unique_id), 128)
        lib().call("mh_comm_init", self.rank, self.world, ctypes.cast(buf, ctypes.c_void_p), self.device, ctypes.byref(self._h))
        r, w, v = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        lib().call("mh_comm_info", self._h, ctypes.byref(r), ctypes.byref(w), ctypes.byref(v))
        assert (r.value, w.value) == (self.rank, self.world)
        self.rccl_version = v.value

    @staticmethod
    def new_unique_id() -> bytes:
        buf = ctypes.create_string_buffer(128)
        lib().call("mh_comm_unique_id", ctypes.cast(buf, ctypes.c_void_p))
        return buf.raw

    @classmethod
    def from_process_group(cls, device: int, group=None) -> "MHComm":
        """rank / world from ``torch.distributed``; rank 0's id travels through the existing process group"""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return cls(0, 1, device)
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [cls.new_unique_id() if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        return cls(rank, world, device, box[0])

    def allreduce_(self, t: torch.Tensor, mean: bool = True, stream: Optional[torch.cuda.Stream] = None) -> torch.Tensor:
        """in place, on ``stream`` (default: torch's current stream): the mean (or sum) of ``t`` over the ranks"""
        assert t.is_cuda and t.is_contiguous()
        st = (stream or torch.cuda.current_stream()).cuda_stream
        lib().call("mh_comm_allreduce", self._h, t.data_ptr(), t.numel(), _dt(t), 1 if mean else 0, st)
        return t

    def broadcast_(self, t: torch.Tensor, root: int = 0, stream: Optional[torch.cuda.Stream] = None) -> torch.Tensor:
        assert t.is_cuda and t.is_contiguous()
        st = (stream or torch.cuda.current_stream()).cuda_stream
        lib().call("mh_comm_broadcast", self._h, t.data_ptr(), t.numel(), _dt(t), int(root), st)
        return t

    def close(self) -> None:
        if self._h is not None and self._h.value:
            lib().call("mh_comm_destroy", self._h)
            self._h = None

    def __del__(self):
        # (not during interpreter teardown: ncclCommDestroy would run against a HIP runtime / librccl that may already be
        #  finalising; the process exit frees the communicator)
        if sys is None or sys.is_finalizing():  # (module globals are cleared during teardown)
            return
        try:
            self.close()
        except Exception:
            pass
