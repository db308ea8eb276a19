"""Thin operator layer: torch CUDA tensors in, C-ABI calls out.

torch supplies device memory and the current stream only; every computation
here is a hand-written HIP kernel inside libdlrm_amd.so.
"""
import ctypes as C

import torch

from . import _capi
from ._capi import (BF16, F32, I32, I64, EmbBagDesc, SparseOpt, check)  # noqa: F401


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _row_dtype(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise TypeError("embedding rows must be float32 or bfloat16, got %s" % t.dtype)


def _idx_dtype(t):
    if t.dtype == torch.int32:
        return I32
    if t.dtype == torch.int64:
        return I64
    raise TypeError("indices must be int32 or int64, got %s" % t.dtype)


def _require_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _capi.DlrmError("dlrm_amd ops need device tensors (there is no CPU fallback)")


def table_init_uniform_(weight, seed, first=0):
    """In-place uniform [0,1) init of a table (embedding_table_init::write)."""
    _require_cuda(weight)
    assert weight.is_contiguous()
    check(_capi.lib().dlrm_table_init_uniform(_ptr(weight), _row_dtype(weight), C.c_int64(weight.numel()),
                                              C.c_uint64(seed), C.c_int64(first), _stream()))
    return weight


def make_embbag_desc(weight, indices, offsets, batch, num_tables=1, table_row_offsets=None,
                     pooling=_capi.POOL_SUM, include_last_offset=False, error_flag=None):
    _require_cuda(weight, indices, offsets, table_row_offsets, error_flag)
    assert weight.dim() == 2 and weight.is_contiguous()
    assert indices.is_contiguous() and offsets.is_contiguous() and indices.dtype == offsets.dtype
    n_off = num_tables * batch + (1 if include_last_offset else 0)
    if offsets.numel() != n_off:
        raise ValueError("offsets has %d entries, expected %d" % (offsets.numel(), n_off))
    if table_row_offsets is not None:
        assert table_row_offsets.dtype == torch.int64 and table_row_offsets.numel() == num_tables + 1
    d = EmbBagDesc()
    d.d_weight = weight.data_ptr()
    d.d_table_row_offsets = 0 if table_row_offsets is None else table_row_offsets.data_ptr()
    d.total_rows = weight.shape[0]
    d.num_tables = num_tables
    d.dim = weight.shape[1]
    d.row_dtype = _row_dtype(weight)
    d.idx_dtype = _idx_dtype(indices)
    d.d_indices = indices.data_ptr()
    d.d_offsets = offsets.data_ptr()
    d.n_indices = indices.numel()
    d.batch = batch
    d.include_last_offset = int(include_last_offset)
    d.pooling = pooling
    d.d_error_flag = 0 if error_flag is None else error_flag.data_ptr()
    return d


def embbag_forward(weight, indices, offsets, batch, num_tables=1, table_row_offsets=None,
                   pooling=_capi.POOL_SUM, include_last_offset=False, out=None, out_ld=None,
                   error_flag=None):
    """Pooled embeddings [batch, T*dim] (or written into `out` with row stride out_ld)."""
    d = make_embbag_desc(weight, indices, offsets, batch, num_tables, table_row_offsets, pooling,
                         include_last_offset, error_flag)
    if out is None:
        out_ld = num_tables * weight.shape[1]
        out = torch.empty((batch, out_ld), dtype=torch.float32, device=weight.device)
    else:
        _require_cuda(out)
        assert out.dtype == torch.float32 and out_ld is not None
    check(_capi.lib().dlrm_embbag_forward(C.byref(d), _ptr(out), C.c_int64(out_ld), _stream()))
    return out


def embbag_backward_workspace_bytes(desc):
    n = C.c_size_t(0)
    check(_capi.lib().dlrm_embbag_backward_workspace_bytes(C.byref(desc), C.byref(n)))
    return int(n.value)


def embbag_backward_update(weight, indices, offsets, batch, grad, num_tables=1, table_row_offsets=None,
                           pooling=_capi.POOL_SUM, include_last_offset=False, optimizer=_capi.OPT_SGD,
                           dup_mode=_capi.DUP_ACCUMULATE, lr=0.01, eps=1e-8, opt_state=None, grad_ld=None,
                           workspace=None, error_flag=None):
    """Sparse backward + fused row-wise SGD/Adagrad; updates `weight` (and
    `opt_state`) in place.  `grad` holds the pooled-vector gradients in the
    forward output layout.  Returns the workspace so it can be reused."""
    d = make_embbag_desc(weight, indices, offsets, batch, num_tables, table_row_offsets, pooling,
                         include_last_offset, error_flag)
    _require_cuda(grad, opt_state, workspace)
    assert grad.dtype == torch.float32
    if grad_ld is None:
        assert grad.dim() == 2 and grad.is_contiguous()
        grad_ld = grad.shape[1]
    need = embbag_backward_workspace_bytes(d)
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(need, dtype=torch.uint8, device=weight.device)
    o = SparseOpt()
    o.optimizer, o.dup_mode, o.lr, o.eps = optimizer, dup_mode, lr, eps
    o.d_opt_state = 0 if opt_state is None else opt_state.data_ptr()
    check(_capi.lib().dlrm_embbag_backward_update(C.byref(d), _ptr(grad), C.c_int64(grad_ld), C.byref(o),
                                                  _ptr(workspace), C.c_size_t(workspace.numel()), _stream()))
    return workspace
