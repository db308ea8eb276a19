"""Recorded train step: the operators of one step are described to libdlrm_amd.so once
(dlrm_plan_add_*) and replayed per batch with one C call per stretch between two collectives
(dlrm_plan_run).  The Python work per launch -- descriptors, stream queries, views -- is paid once,
which matters when a rank's share of the batch is small (8 GPUs: 8192 samples, about a millisecond
of kernels).

    rec = StepRecorder(indices, offsets)     # the batch the step is traced with
    with rec:
        first = rec.size; model_code_using_dlrm_amd_ops(); last = rec.size
    plan = rec.plan
    plan.set_batch(indices, offsets); plan.run(first, last)
"""
import ctypes as C

import torch

from . import _capi, ops
from ._capi import check

OP_NAMES = ("embbag_forward", "embbag_backward_update", "linear_forward", "linear_backward_data",
            "linear_backward_weight", "sgd_step", "interact_dot_forward", "interact_dot_backward", "loss", "copy_2d",
            "embbag_backward_prepare", "embbag_backward_apply")


class StepPlan:
    def __init__(self):
        self.L = _capi.lib()
        self.h = C.c_void_p()
        check(self.L.dlrm_plan_create(C.byref(self.h)))
        self.meta = []          # per entry: (public operator name, args, kwargs) it was recorded from
        self.keep = []          # tensors the recorded pointers refer to

    def __del__(self):
        try:
            if self.h:
                self.L.dlrm_plan_destroy(self.h)
                self.h = None
        except Exception:
            pass

    @property
    def size(self):
        return int(self.L.dlrm_plan_size(self.h))

    def op_name(self, entry):
        return OP_NAMES[int(self.L.dlrm_plan_op(self.h, entry))]

    def set_batch(self, indices, offsets):
        check(self.L.dlrm_plan_set_batch(self.h, C.c_void_p(indices.data_ptr()), C.c_void_p(offsets.data_ptr()),
                                         C.c_int64(indices.numel())))

    def run(self, first, last, stream=None):
        s = torch.cuda.current_stream() if stream is None else stream
        check(self.L.dlrm_plan_run(self.h, first, last, C.c_void_p(s.cuda_stream)))

    def enable_timing(self, max_runs):
        check(self.L.dlrm_plan_enable_timing(self.h, int(max_runs)))

    def collect_timing(self):
        """(ms summed per entry, replays per entry) since the last call; waits for the events."""
        n = self.size
        ms = (C.c_double * n)()
        runs = (C.c_int32 * n)()
        check(self.L.dlrm_plan_collect_timing(self.h, ms, runs))
        return list(ms), list(runs)


class StepRecorder:
    """Installed as dlrm_amd.ops._recorder while a step is traced: the operators hand their C-ABI calls
    over instead of launching them."""

    def __init__(self, indices, offsets):
        self.plan = StepPlan()
        self.indices, self.offsets = indices, offsets
        self.plan.keep += [indices, offsets]
        self._call = None

    def __enter__(self):
        if ops._recorder is not None:
            raise RuntimeError("a step is already being recorded")
        ops._recorder = self
        return self

    def __exit__(self, *exc):
        ops._recorder = None
        return False

    @property
    def size(self):
        return self.plan.size

    def note(self, name, args, kwargs):
        self._call = (name, args, kwargs)
        for a in list(args) + list(kwargs.values()):
            if isinstance(a, torch.Tensor):
                self.plan.keep.append(a)

    def add(self, name, args):
        L = self.plan.L
        cargs = []
        for a in args:
            if isinstance(a, C.Structure):
                self.plan.keep.append(a)
                cargs.append(C.byref(a))
            else:
                cargs.append(a)
        if name in ("embbag_forward", "embbag_backward_update", "embbag_backward_prepare", "embbag_backward_apply"):
            d = args[0]
            if d.d_indices != self.indices.data_ptr():
                raise RuntimeError("a recorded EmbeddingBag call must use the batch's index tensor")
            isz = self.offsets.element_size()
            rel = d.d_offsets - self.offsets.data_ptr()
            if rel < 0 or rel % isz or rel // isz >= self.offsets.numel():
                raise RuntimeError("a recorded EmbeddingBag call must use a slice of the batch's offsets")
            cargs.insert(1, C.c_int64(rel // isz))
        rc = getattr(L, "dlrm_plan_add_" + name)(self.plan.h, *cargs)
        if rc < 0:
            raise _capi.DlrmError("dlrm_plan_add_%s: %s" % (name, L.dlrm_last_error().decode()))
        self.plan.meta.append(self._call)
        return rc
