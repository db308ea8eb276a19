"""Thin operator layer: torch CUDA tensors in, C-ABI calls out.

torch supplies device memory and the current stream only; every computation
here is a hand-written HIP kernel inside libdlrm_amd.so.
"""
import ctypes as C
import os

import torch

from . import _capi
from ._capi import (BF16, F32, I32, I64, EmbBagDesc, SparseOpt, check)  # noqa: F401


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


# While a StepRecorder is installed (dlrm_amd.plan), the operators below do everything except launch:
# their C-ABI call is handed to the recorder, which appends it to a dlrm_plan.
_recorder = None
_gemm_mode = {"0": 0, "1": 1, "2": 2}.get(os.environ.get("DLRM_GEMM_MODE", "1").strip(), 1)


def _emit(name, *args):
    """Launch `dlrm_<name>(*args, stream)` now, or record it."""
    if _recorder is None:
        args = [C.byref(a) if isinstance(a, C.Structure) else a for a in args]
        check(getattr(_capi.lib(), "dlrm_" + name)(*args, _stream()))
    else:
        _recorder.add(name, args)


def _op(fn):
    """Marks a public operator: a recorder learns which operator (and with which arguments) the
    entries recorded next belong to."""
    import functools

    @functools.wraps(fn)
    def wrapped(*a, **kw):
        if _recorder is not None:
            _recorder.note(fn.__name__, a, kw)
        return fn(*a, **kw)
    return wrapped


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
                     pooling=_capi.POOL_SUM, include_last_offset=False, error_flag=None, shard_row_begin=None):
    _require_cuda(weight, indices, offsets, table_row_offsets, error_flag, shard_row_begin)
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
    if shard_row_begin is not None:
        assert shard_row_begin.dtype == torch.int64 and shard_row_begin.numel() == num_tables
        assert shard_row_begin.is_contiguous()
    d.d_shard_row_begin = 0 if shard_row_begin is None else shard_row_begin.data_ptr()
    return d


@_op
def embbag_forward(weight, indices, offsets, batch, num_tables=1, table_row_offsets=None,
                   pooling=_capi.POOL_SUM, include_last_offset=False, out=None, out_ld=None,
                   error_flag=None, shard_row_begin=None):
    """Pooled embeddings [batch, T*dim] (or written into `out` with row stride out_ld).
    shard_row_begin (int64[T]): the tables are row-wise shards starting at these global rows;
    indices stay global and those of other shards are skipped."""
    d = make_embbag_desc(weight, indices, offsets, batch, num_tables, table_row_offsets, pooling,
                         include_last_offset, error_flag, shard_row_begin)
    if out is None:
        out_ld = num_tables * weight.shape[1]
        out = torch.empty((batch, out_ld), dtype=torch.float32, device=weight.device)
    else:
        _require_cuda(out)
        assert out.dtype == torch.float32 and out_ld is not None
    _emit("embbag_forward", d, _ptr(out), C.c_int64(out_ld))
    return out


def embbag_backward_workspace_bytes(desc):
    n = C.c_size_t(0)
    check(_capi.lib().dlrm_embbag_backward_workspace_bytes(C.byref(desc), C.byref(n)))
    return int(n.value)


@_op
def embbag_backward_update(weight, indices, offsets, batch, grad, num_tables=1, table_row_offsets=None,
                           pooling=_capi.POOL_SUM, include_last_offset=False, optimizer=_capi.OPT_SGD,
                           dup_mode=_capi.DUP_ACCUMULATE, lr=0.01, eps=1e-8, opt_state=None, grad_ld=None,
                           workspace=None, error_flag=None, shard_row_begin=None):
    """Sparse backward + fused row-wise SGD/Adagrad; updates `weight` (and
    `opt_state`) in place.  `grad` holds the pooled-vector gradients in the
    forward output layout.  Returns the workspace so it can be reused."""
    d = make_embbag_desc(weight, indices, offsets, batch, num_tables, table_row_offsets, pooling,
                         include_last_offset, error_flag, shard_row_begin)
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
    _emit("embbag_backward_update", d, _ptr(grad), C.c_int64(grad_ld), o, _ptr(workspace),
          C.c_size_t(workspace.numel()))
    return workspace


@_op
def embbag_backward_prepare(weight, indices, offsets, batch, num_tables=1, table_row_offsets=None,
                            pooling=_capi.POOL_SUM, include_last_offset=False, workspace=None, error_flag=None,
                            shard_row_begin=None):
    """First half of embbag_backward_update: occurrences sorted by row into the workspace (needs the ids
    only).  Returns the workspace that embbag_backward_apply must be given."""
    d = make_embbag_desc(weight, indices, offsets, batch, num_tables, table_row_offsets, pooling,
                         include_last_offset, error_flag, shard_row_begin)
    need = embbag_backward_workspace_bytes(d)
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(need, dtype=torch.uint8, device=weight.device)
    _emit("embbag_backward_prepare", d, _ptr(workspace), C.c_size_t(workspace.numel()))
    return workspace


@_op
def embbag_backward_apply(weight, indices, offsets, batch, grad, workspace, num_tables=1, table_row_offsets=None,
                          pooling=_capi.POOL_SUM, include_last_offset=False, optimizer=_capi.OPT_SGD,
                          dup_mode=_capi.DUP_ACCUMULATE, lr=0.01, eps=1e-8, opt_state=None, grad_ld=None,
                          error_flag=None, shard_row_begin=None):
    """Second half: reduce the gradients over the prepared order and update the rows."""
    d = make_embbag_desc(weight, indices, offsets, batch, num_tables, table_row_offsets, pooling,
                         include_last_offset, error_flag, shard_row_begin)
    _require_cuda(grad, opt_state, workspace)
    assert grad.dtype == torch.float32
    if grad_ld is None:
        assert grad.dim() == 2 and grad.is_contiguous()
        grad_ld = grad.shape[1]
    o = SparseOpt()
    o.optimizer, o.dup_mode, o.lr, o.eps = optimizer, dup_mode, lr, eps
    o.d_opt_state = 0 if opt_state is None else opt_state.data_ptr()
    _emit("embbag_backward_apply", d, _ptr(grad), C.c_int64(grad_ld), o, _ptr(workspace),
          C.c_size_t(workspace.numel()))
    return workspace


# ------------------------------------------------------------------ dense


def _ld(t):
    assert t.dim() == 2 and t.stride(1) == 1, "need a row-major 2-D view"
    return t.stride(0)


class SplitWeights:
    """The f16 planes of a weight matrix as dlrm_split_weights leaves them (f16x2 evaluation): keeps the device
    buffers and the host descriptor alive; pass it as w_planes= to linear_forward / linear_backward_data together with
    w_absmax=self.absmax.  Split again after the matrix changes."""

    def __init__(self, w):
        _require_cuda(w)
        N, K = w.shape
        nbytes = C.c_size_t()
        check(_capi.lib().dlrm_weight_planes_bytes(N, K, C.byref(nbytes)))
        self.w = w
        self.absmax = torch.zeros(1, dtype=torch.float32, device=w.device)
        self.planes = torch.empty(nbytes.value, dtype=torch.uint8, device=w.device)
        self.item = _capi.WeightPlanes(w.data_ptr(), _ld(w), N, K, self.absmax.data_ptr(), self.planes.data_ptr())
        self.split()

    def split(self):
        self.absmax.zero_()
        check(_capi.lib().dlrm_absmax(C.c_void_p(self.w.data_ptr()), C.c_int64(self.w.shape[0]), C.c_int32(self.w.shape[1]),
                                      C.c_int64(_ld(self.w)), C.c_void_p(self.absmax.data_ptr()), _stream()))
        check(_capi.lib().dlrm_split_weights(C.byref(self.item), 1, _stream()))

    def ptr(self):
        return C.addressof(self.item)


@_op
def linear_forward(x, w, bias, relu, out=None, x_absmax=None, w_absmax=None, y_absmax=None, w_planes=None):
    """y = act(x W^T + b).  x [M,K] and w [N,K] may be column slices of wider
    (padded) buffers: the row stride is passed as the leading dimension.
    *_absmax (f16x2 mode): one-element float32 tensors holding max |x|, max |W| (None: measured by the call) and
    receiving max |y| (atomicMax: zero it first).  w_planes (f16x2 mode): a SplitWeights of w."""
    _require_cuda(x, w, bias, out)
    M, K = x.shape
    N = w.shape[0]
    assert w.shape[1] == K and x.dtype == torch.float32 and w.dtype == torch.float32
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=x.device)
    d = _capi.LinearDesc()
    d.M, d.N, d.K = M, N, K
    d.d_x, d.x_ld = x.data_ptr(), _ld(x)
    d.d_w, d.w_ld = w.data_ptr(), _ld(w)
    d.d_bias = 0 if bias is None else bias.data_ptr()
    d.d_y, d.y_ld = out.data_ptr(), _ld(out)
    d.act = _capi.ACT_RELU if relu else _capi.ACT_NONE
    d.gemm_mode = _gemm_field()
    d.d_x_absmax, d.d_w_absmax, d.d_y_absmax = _ptr0(x_absmax), _ptr0(w_absmax), _ptr0(y_absmax)
    if w_planes is not None:
        d.w_planes, d.d_w_absmax = w_planes.ptr(), w_planes.absmax.data_ptr()
    _emit("linear_forward", d)
    return out


@_op
def linear_backward_data(dz, w, relu_mask=None, out=None, dz_absmax=None, w_absmax=None, dx_absmax=None, w_planes=None):
    """dx = dz W, optionally masked with (relu_mask > 0).  *_absmax, w_planes: see linear_forward."""
    _require_cuda(dz, w, relu_mask, out)
    M, N = dz.shape
    K = w.shape[1]
    if out is None:
        out = torch.empty((M, K), dtype=torch.float32, device=dz.device)
    d = _capi.LinearBwdDesc()
    d.M, d.N, d.K = M, N, K
    d.d_dz, d.dz_ld = dz.data_ptr(), _ld(dz)
    d.d_w, d.w_ld = w.data_ptr(), _ld(w)
    d.d_dx, d.dx_ld = out.data_ptr(), _ld(out)
    if relu_mask is not None:
        d.d_relu_mask, d.relu_mask_ld = relu_mask.data_ptr(), _ld(relu_mask)
    d.gemm_mode = _gemm_field()
    d.d_dz_absmax, d.d_w_absmax, d.d_dx_absmax = _ptr0(dz_absmax), _ptr0(w_absmax), _ptr0(dx_absmax)
    if w_planes is not None:
        d.w_planes, d.d_w_absmax = w_planes.ptr(), w_planes.absmax.data_ptr()
    _emit("linear_backward_data", d)
    return out


def _ptr0(t):
    return 0 if t is None else t.data_ptr()


def absmax_(x, slot):
    """max |x| of a 2-D float32 tensor (rows may be strided) folded into the one-element tensor `slot` (atomic maximum:
    zero it to start a new measurement)."""
    _require_cuda(x, slot)
    assert x.dim() == 2 and x.dtype == torch.float32 and slot.dtype == torch.float32 and slot.numel() == 1
    check(_capi.lib().dlrm_absmax(_ptr(x), C.c_int64(x.shape[0]), C.c_int32(x.shape[1]), C.c_int64(_ld(x)), _ptr(slot), _stream()))
    return slot


def linear_backward_weight_workspace_bytes(M, N, K):
    n = C.c_size_t(0)
    check(_capi.lib().dlrm_linear_backward_weight_workspace_bytes(C.c_int64(M), N, K, C.byref(n)))
    return int(n.value)


@_op
def linear_backward_weight(dz, x, dw=None, db=None, workspace=None, want_db=True, dz_absmax=None, x_absmax=None):
    """dW = dz^T x [N,K], db = column sums of dz.  *_absmax: see linear_forward."""
    _require_cuda(dz, x, dw, db, workspace)
    M, N = dz.shape
    K = x.shape[1]
    if dw is None:
        dw = torch.empty((N, K), dtype=torch.float32, device=dz.device)
    if db is None and want_db:
        db = torch.empty((N,), dtype=torch.float32, device=dz.device)
    need = linear_backward_weight_workspace_bytes(M, N, K)
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(need, dtype=torch.uint8, device=dz.device)
    d = _capi.LinearBwdDesc()
    d.M, d.N, d.K = M, N, K
    d.d_dz, d.dz_ld = dz.data_ptr(), _ld(dz)
    d.d_x, d.x_ld = x.data_ptr(), _ld(x)
    d.d_dw, d.dw_ld = dw.data_ptr(), _ld(dw)
    d.d_db = 0 if db is None else db.data_ptr()
    d.gemm_mode = _gemm_field()
    d.d_dz_absmax, d.d_x_absmax = _ptr0(dz_absmax), _ptr0(x_absmax)
    _emit("linear_backward_weight", d, _ptr(workspace), C.c_size_t(workspace.numel()))
    return dw, db, workspace


def linear_backward_weight_static(dz, x, dw, db=None):
    """linear_backward_weight through the entry point that allocates nothing: partial sums go to the library's
    static device array (layers beyond its size are refused); outputs must be given."""
    _require_cuda(dz, x, dw, db)
    M, N = dz.shape
    K = x.shape[1]
    d = _capi.LinearBwdDesc()
    d.M, d.N, d.K = M, N, K
    d.d_dz, d.dz_ld = dz.data_ptr(), _ld(dz)
    d.d_x, d.x_ld = x.data_ptr(), _ld(x)
    d.d_dw, d.dw_ld = dw.data_ptr(), _ld(dw)
    d.d_db = 0 if db is None else db.data_ptr()
    d.gemm_mode = _gemm_field()
    check(_capi.lib().dlrm_linear_backward_weight_static(C.byref(d), _stream()))
    return dw, db


@_op
def sgd_step_(w, g, lr):
    _require_cuda(w, g)
    assert w.is_contiguous() and g.is_contiguous() and w.numel() == g.numel()
    _emit("sgd_step", _ptr(w), _ptr(g), C.c_int64(w.numel()), C.c_float(lr))
    return w


# ------------------------------------------------- interaction and losses


@_op
def interact_dot_forward(x, out=None, out_ld=None):
    """x [B,F,D] -> [B, D + F(F-1)/2] (row stride out_ld, tail zero-filled)."""
    _require_cuda(x, out)
    B, F, D = x.shape
    assert x.is_contiguous()
    n_out = D + F * (F - 1) // 2
    if out is None:
        out_ld = n_out if out_ld is None else out_ld
        out = torch.empty((B, out_ld), dtype=torch.float32, device=x.device)
    _emit("interact_dot_forward", _ptr(x), C.c_int64(B), C.c_int32(F), C.c_int32(D), _ptr(out), C.c_int64(out_ld))
    return out


@_op
def interact_dot_backward(x, dout, relu_mask_feature0=False, out=None):
    _require_cuda(x, dout, out)
    B, F, D = x.shape
    if out is None:
        out = torch.empty_like(x)
    _emit("interact_dot_backward", _ptr(x), _ptr(dout), C.c_int64(_ld(dout)), C.c_int64(B), C.c_int32(F), C.c_int32(D),
          _ptr(out), C.c_int32(int(relu_mask_feature0)))
    return out


def interact_dot_forward_gather(weight, indices, offsets, batch, num_tables, table_row_offsets, x0, out=None, out_ld=None,
                                error_flag=None, include_last_offset=False, x_copy=None):
    """interact_dot_forward on x = [x0 | one row per table], the rows read straight from the tables (every bag of the
    batch holds exactly one id).  x0: [B, D] row-major view (feature 0)."""
    d = make_embbag_desc(weight, indices, offsets, batch, num_tables, table_row_offsets, _capi.POOL_SUM,
                         include_last_offset, error_flag)
    _require_cuda(x0, out)
    F, D = num_tables + 1, weight.shape[1]
    n_out = D + F * (F - 1) // 2
    if out is None:
        out_ld = n_out if out_ld is None else out_ld
        out = torch.empty((batch, out_ld), dtype=torch.float32, device=weight.device)
    if x_copy is not None:
        _require_cuda(x_copy)
        assert x_copy.is_contiguous() and x_copy.numel() == batch * F * D and x_copy.dtype == torch.float32
    check(_capi.lib().dlrm_interact_dot_forward_gather(C.byref(d), _ptr(x0), C.c_int64(_ld(x0)), _ptr(out), C.c_int64(out_ld),
                                                       _ptr(x_copy) if x_copy is not None else None, _stream()))
    return out


def interact_dot_backward_gather(weight, indices, offsets, batch, num_tables, table_row_offsets, x0, dout, relu_mask_feature0=False,
                                 out=None, error_flag=None, include_last_offset=False):
    """interact_dot_backward on x = [x0 | one row per table] with the rows read from the tables again (the backward
    pass of interact_dot_forward_gather called without x_copy).  Returns dx [B, F, D]."""
    d = make_embbag_desc(weight, indices, offsets, batch, num_tables, table_row_offsets, _capi.POOL_SUM,
                         include_last_offset, error_flag)
    _require_cuda(x0, dout, out)
    F, D = num_tables + 1, weight.shape[1]
    if out is None:
        out = torch.empty((batch, F, D), dtype=torch.float32, device=weight.device)
    assert out.is_contiguous() and out.numel() == batch * F * D and out.dtype == torch.float32
    check(_capi.lib().dlrm_interact_dot_backward_gather(C.byref(d), _ptr(x0), C.c_int64(_ld(x0)), _ptr(dout), C.c_int64(_ld(dout)),
                                                        _ptr(out), C.c_int32(int(relu_mask_feature0)), _stream()))
    return out


def _views(ts, D):
    v = (_capi.FeatureView * len(ts))()
    for it, t in zip(v, ts):
        _require_cuda(t)
        assert t.dim() == 2 and t.shape[1] == D and t.stride(1) == 1 and t.dtype == torch.float32
        it.d_base, it.stride = t.data_ptr(), t.stride(0)
    return v


def interact_dot_forward_views(xs, out=None, out_ld=None):
    """interact_dot_forward on features given one by one: xs[f] is a [B, D] row-major view (any row stride)."""
    B, D = xs[0].shape
    F = len(xs)
    n_out = D + F * (F - 1) // 2
    if out is None:
        out_ld = n_out if out_ld is None else out_ld
        out = torch.empty((B, out_ld), dtype=torch.float32, device=xs[0].device)
    check(_capi.lib().dlrm_interact_dot_forward_views(_views(xs, D), C.c_int64(B), C.c_int32(F), C.c_int32(D), _ptr(out),
                                                      C.c_int64(out_ld), _stream()))
    return out


def interact_dot_backward_views(xs, dout, dxs, relu_mask_feature0=False):
    """interact_dot_backward likewise: the gradient of feature f is written to the [B, D] view dxs[f]."""
    B, D = xs[0].shape
    F = len(xs)
    _require_cuda(dout)
    check(_capi.lib().dlrm_interact_dot_backward_views(_views(xs, D), _ptr(dout), C.c_int64(_ld(dout)), C.c_int64(B), C.c_int32(F),
                                                       C.c_int32(D), _views(dxs, D), C.c_int32(int(relu_mask_feature0)), _stream()))
    return dxs


@_op
def loss_mse(pred, target, relu_mask_pred=False, loss=None, dpred=None, mean_over=None, want_loss=True, want_grad=True):
    """want_loss / want_grad = False: that output is not computed (None is returned in its place)."""
    _require_cuda(pred, target)
    n = pred.numel()
    assert pred.is_contiguous() and target.is_contiguous() and target.numel() == n
    loss = (torch.empty(1, dtype=torch.float32, device=pred.device) if loss is None else loss) if want_loss else None
    dpred = (torch.empty_like(pred) if dpred is None else dpred) if want_grad else None
    _emit("loss_mse", _ptr(pred), _ptr(target), C.c_int64(n), C.c_int64(mean_over or n), _ptr(loss), _ptr(dpred),
          C.c_int32(int(relu_mask_pred)))
    return loss, dpred


@_op
def loss_bce_logits(logit, target, loss=None, dlogit=None, mean_over=None, want_loss=True, want_grad=True):
    _require_cuda(logit, target)
    n = logit.numel()
    assert logit.is_contiguous() and target.is_contiguous() and target.numel() == n
    loss = (torch.empty(1, dtype=torch.float32, device=logit.device) if loss is None else loss) if want_loss else None
    dlogit = (torch.empty_like(logit) if dlogit is None else dlogit) if want_grad else None
    _emit("loss_bce_logits", _ptr(logit), _ptr(target), C.c_int64(n), C.c_int64(mean_over or n), _ptr(loss),
          _ptr(dlogit))
    return loss, dlogit


def copy_2d_many_(pairs):
    """dst[:, :] = src[:, :] for every (dst, src) pair of equally shaped row-major float32 views, one launch per 16 pairs."""
    items = (_capi.CopyItem * len(pairs))()
    for it, (dst, src) in zip(items, pairs):
        _require_cuda(dst, src)
        assert dst.shape == src.shape and dst.dtype == torch.float32 and src.dtype == torch.float32
        it.d_dst, it.dst_ld, it.d_src, it.src_ld = dst.data_ptr(), _ld(dst), src.data_ptr(), _ld(src)
        it.rows, it.cols = dst.shape
    check(_capi.lib().dlrm_copy_2d_many(items, len(pairs), _stream()))


@_op
def copy_2d_(dst, src):
    """dst[:, :] = src[:, :] for two row-major 2-D float32 views (any row strides)."""
    _require_cuda(dst, src)
    assert dst.shape == src.shape and dst.dtype == torch.float32 and src.dtype == torch.float32
    _emit("copy_2d", _ptr(dst), C.c_int64(_ld(dst)), _ptr(src), C.c_int64(_ld(src)), C.c_int64(dst.shape[0]),
          C.c_int64(dst.shape[1]))
    return dst


def ids_regroup(recv, world, local_batch, n_tables, hot_offsets=None, offset_base=0, indices=None, offsets=None):
    """After the index exchange: recv = [rank][table][sample][hot] ids -> (indices table-major over the global
    batch, CSR offsets raised by offset_base).  hot_offsets: device int32[n_tables + 1] or None (one id per bag)."""
    _require_cuda(recv, hot_offsets, indices, offsets)
    Bg = world * local_batch
    if indices is None:
        indices = torch.empty(recv.numel(), dtype=recv.dtype, device=recv.device)
    if offsets is None:
        offsets = torch.empty(n_tables * Bg + 1, dtype=recv.dtype, device=recv.device)
    per_sample = recv.numel() // Bg if Bg else n_tables      # the sum of these tables' hotness
    check(_capi.lib().dlrm_ids_regroup(_ptr(recv), C.c_int32(world), C.c_int64(local_batch), C.c_int32(n_tables),
                                       _ptr(hot_offsets), C.c_int64(per_sample), C.c_int64(offset_base), _ptr(indices),
                                       _ptr(offsets), _idx_dtype(recv), _stream()))
    return indices, offsets


def set_gemm_mode(mode):
    """How the linear operators issued from now on evaluate their float32 products (a field of every layer
    descriptor, so a recorded step keeps the mode it was recorded with): 0 float32 MFMA
    (v_mfma_f32_32x32x2_f32); 1 exact bf16x3 split products on the bf16 MFMA with float32 accumulation
    (float32 accuracy, 8/3 of the matrix-pipe rate; the default); 2 two-piece f16 splits of the operands scaled by
    a power of two (three cross terms, representation error near one float32 ulp).  Environment: DLRM_GEMM_MODE=0|1|2."""
    global _gemm_mode
    if int(mode) not in (0, 1, 2):
        raise ValueError("gemm mode must be 0 (float32 MFMA), 1 (bf16x3 split) or 2 (f16x2 split)")
    _gemm_mode = int(mode)


def get_gemm_mode():
    return _gemm_mode


def _gemm_field():
    return (_capi.GEMM_F32_MFMA, _capi.GEMM_BF16X3, _capi.GEMM_F16X2)[_gemm_mode]
