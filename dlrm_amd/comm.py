"""ctypes mirror of the C ABI's collectives (dlrm_comm_*, RCCL): what a C++ host of the sharded step calls.
The Python model itself uses torch.distributed (the same RCCL underneath)."""
import ctypes as C

import torch

from . import _capi
from ._capi import check

ID_BYTES = 128


def unique_id():
    """Bytes that rank 0 hands to every other rank (any channel) before Comm(...)."""
    buf = (C.c_uint8 * ID_BYTES)()
    check(_capi.lib().dlrm_comm_unique_id(buf))
    return bytes(buf)


def unique_id_rehearsal():
    """An id for the rehearsal transport (ranks of one machine exchanging through shared host memory): lets several
    ranks share one GPU, which RCCL refuses."""
    buf = (C.c_uint8 * ID_BYTES)()
    check(_capi.lib().dlrm_comm_unique_id_rehearsal(buf))
    return bytes(buf)


class Comm:
    def __init__(self, comm_id, rank, world):
        self.L = _capi.lib()
        self.rank, self.world = rank, world
        buf = (C.c_uint8 * ID_BYTES)(*comm_id)
        self.h = C.c_void_p()
        check(self.L.dlrm_comm_init(buf, rank, world, C.byref(self.h)))

    def close(self):
        if getattr(self, "h", None):
            torch.cuda.synchronize()
            check(self.L.dlrm_comm_destroy(self.h))
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _stream():
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    @staticmethod
    def _f32(t):
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
        return C.c_void_p(t.data_ptr())

    def all_reduce_sum_(self, t):
        check(self.L.dlrm_comm_all_reduce_sum(self.h, self._f32(t), C.c_int64(t.numel()), self._stream()))
        return t

    def all_to_all(self, recv, send, recv_counts, send_counts):
        sc = (C.c_int64 * self.world)(*[int(x) for x in send_counts])
        rc = (C.c_int64 * self.world)(*[int(x) for x in recv_counts])
        assert sum(send_counts) <= send.numel() and sum(recv_counts) <= recv.numel()
        check(self.L.dlrm_comm_all_to_all(self.h, self._f32(send), sc, self._f32(recv), rc, self._stream()))
        return recv

    def reduce_scatter_sum(self, recv, send):
        assert send.numel() == recv.numel() * self.world
        check(self.L.dlrm_comm_reduce_scatter_sum(self.h, self._f32(send), self._f32(recv), C.c_int64(recv.numel()),
                                                  self._stream()))
        return recv

    def all_gather(self, recv, send):
        assert recv.numel() == send.numel() * self.world
        check(self.L.dlrm_comm_all_gather(self.h, self._f32(send), self._f32(recv), C.c_int64(send.numel()), self._stream()))
        return recv
