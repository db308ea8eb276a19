"""TEST-ONLY stand-ins for dlrm_amd.ops built on the CPU oracle, so that the
multi-rank logic of dlrm_amd.model (sharding plan, all-to-all layout, gradient
all-reduce, loss scaling) can run on CPU tensors under the gloo backend.  The
product never imports this module; it is installed by monkeypatching in
tests/test_distributed_cpu.py only."""
import numpy as np
import torch

from oracle import oracle as orc


def _np(t):
    return t.detach().numpy()


def _write2d(dst, arr):
    dst.copy_(torch.from_numpy(np.ascontiguousarray(arr)).view(dst.shape))


def table_init_uniform_(weight, seed, first=0):
    weight.copy_(torch.from_numpy(orc.table_init_uniform(weight.numel(), seed, first)).view(weight.shape))
    return weight


def _shard_filter(indices, offsets, batch, num_tables, tro, shard_row_begin):
    """Rows of a row-wise shard: drop the ids of other shards, re-base the rest.  Returns the
    filtered CSR and the full bag lengths (mean pooling divides by those)."""
    idx, off = _np(indices), _np(offsets)
    tro = _np(tro) if tro is not None else None
    beg = _np(shard_row_begin)
    n_bags = num_tables * batch
    ends = np.append(off[1:n_bags], off[n_bags] if len(off) > n_bags else len(idx)) if n_bags else np.zeros(0, off.dtype)
    new_idx, new_off, lens = [], [0], []
    for g in range(n_bags):
        t = g // batch
        rows_t = tro[t + 1] - tro[t]
        seg = idx[off[g]:ends[g]] - beg[t]
        keep = seg[(seg >= 0) & (seg < rows_t)]
        new_idx.append(keep)
        new_off.append(new_off[-1] + len(keep))
        lens.append(ends[g] - off[g])
    new_idx = np.concatenate(new_idx).astype(idx.dtype) if new_idx else np.zeros(0, idx.dtype)
    return new_idx, np.asarray(new_off, dtype=off.dtype), np.asarray(lens, dtype=np.int64)


def embbag_forward(weight, indices, offsets, batch, num_tables=1, table_row_offsets=None, pooling=0,
                   include_last_offset=False, out=None, out_ld=None, error_flag=None, shard_row_begin=None):
    D = weight.shape[1]
    if shard_row_begin is not None:
        assert include_last_offset
        fi, fo, lens = _shard_filter(indices, offsets, batch, num_tables, table_row_offsets, shard_row_begin)
        res = orc.embbag_forward(_np(weight), fi, fo, batch, num_tables, _np(table_row_offsets), 0, True)
        if pooling == 1:
            sc = np.where(lens > 0, 1.0 / np.maximum(lens, 1), 0.0).astype(np.float32)
            res = (res.reshape(batch, num_tables, D) * sc.reshape(num_tables, batch).T[:, :, None]).reshape(batch, -1)
            res = res.astype(np.float32)
    else:
        res = orc.embbag_forward(_np(weight), _np(indices), _np(offsets), batch, num_tables,
                                 None if table_row_offsets is None else _np(table_row_offsets), pooling,
                                 include_last_offset)
    flat = out.reshape(-1) if out.is_contiguous() else None
    if out.dim() == 2:
        out[:, :num_tables * D] = torch.from_numpy(res)
    else:
        view = torch.as_strided(flat, (batch, num_tables * D), (out_ld, 1))
        view.copy_(torch.from_numpy(res))
    return out


def embbag_backward_update(weight, indices, offsets, batch, grad, num_tables=1, table_row_offsets=None, pooling=0,
                           include_last_offset=False, optimizer=0, dup_mode=0, lr=0.01, eps=1e-8, opt_state=None,
                           grad_ld=None, workspace=None, error_flag=None, shard_row_begin=None):
    D = weight.shape[1]
    if grad.dim() == 1:
        g = torch.as_strided(grad, (batch, num_tables * D), (grad_ld, 1))
    else:
        g = grad[:, :num_tables * D]
    w = _np(weight)
    st = None if opt_state is None else _np(opt_state)
    gn = np.ascontiguousarray(_np(g))
    if shard_row_begin is not None:
        assert include_last_offset
        fi, fo, lens = _shard_filter(indices, offsets, batch, num_tables, table_row_offsets, shard_row_begin)
        if pooling == 1:   # mean: the row gradient is grad / full bag length; feed it as a sum-pooled gradient
            sc = np.where(lens > 0, 1.0 / np.maximum(lens, 1), 0.0).astype(np.float32)
            gn = (gn.reshape(batch, num_tables, D) * sc.reshape(num_tables, batch).T[:, :, None]).reshape(batch, -1)
            gn = np.ascontiguousarray(gn.astype(np.float32))
        orc.embbag_backward_update(w, fi, fo, batch, gn, num_tables, _np(table_row_offsets), 0, True, optimizer,
                                   dup_mode, lr, eps, st)
        return workspace
    orc.embbag_backward_update(w, _np(indices), _np(offsets), batch, gn, num_tables,
                               None if table_row_offsets is None else _np(table_row_offsets), pooling,
                               include_last_offset, optimizer, dup_mode, lr, eps, st)
    return workspace


def linear_forward(x, w, bias, relu, out=None):
    y = orc.linear_forward(_np(x), _np(w), None if bias is None else _np(bias), relu)
    out.copy_(torch.from_numpy(y))
    return out


def linear_backward_data(dz, w, relu_mask=None, out=None):
    dzn, wn = np.ascontiguousarray(_np(dz)), np.ascontiguousarray(_np(w))
    M, N = dzn.shape
    K = wn.shape[1]
    dx, _, _ = orc.linear_backward(np.zeros((M, K), np.float32), wn, np.ones((M, N), np.float32), dzn, False)
    if relu_mask is not None:
        dx = dx * (_np(relu_mask) > 0)
    out.copy_(torch.from_numpy(dx.astype(np.float32)))
    return out


def linear_backward_weight(dz, x, dw=None, db=None, workspace=None, want_db=True):
    dzn, xn = np.ascontiguousarray(_np(dz)), np.ascontiguousarray(_np(x))
    M, N = dzn.shape
    K = xn.shape[1]
    _, dwn, dbn = orc.linear_backward(xn, np.zeros((N, K), np.float32), np.ones((M, N), np.float32), dzn, False,
                                      need_dx=False)
    dw.copy_(torch.from_numpy(dwn))
    if db is not None:
        db.copy_(torch.from_numpy(dbn))
    return dw, db, workspace


def sgd_step_(w, g, lr):
    wn = _np(w)
    orc.sgd_step(wn, _np(g), lr)
    return w


def interact_dot_forward(x, out=None, out_ld=None):
    res = orc.interact_dot_forward(_np(x))
    out.zero_()
    out[:, :res.shape[1]] = torch.from_numpy(res)
    return out


def interact_dot_backward(x, dout, relu_mask_feature0=False, out=None):
    xn = np.ascontiguousarray(_np(x))
    dx = orc.interact_dot_backward(xn, np.ascontiguousarray(_np(dout)))
    if relu_mask_feature0:
        dx[:, 0, :] *= (xn[:, 0, :] > 0)
    out.copy_(torch.from_numpy(dx))
    return out


def loss_mse(pred, target, relu_mask_pred=False, loss=None, dpred=None, mean_over=None):
    n = pred.numel()
    l, d = orc.mse_loss(_np(pred), _np(target))
    s = n / float(mean_over or n)
    if relu_mask_pred:
        d = d * (_np(pred) > 0)
    loss.fill_(l * s)
    dpred.copy_(torch.from_numpy((d * s).astype(np.float32)))
    return loss, dpred


def loss_bce_logits(logit, target, loss=None, dlogit=None, mean_over=None):
    n = logit.numel()
    l, d = orc.bce_logits_loss(_np(logit), _np(target))
    s = n / float(mean_over or n)
    loss.fill_(l * s)
    dlogit.copy_(torch.from_numpy((d * s).astype(np.float32)))
    return loss, dlogit


def copy_2d_(dst, src):
    dst.copy_(src)
    return dst


def ids_regroup(recv, world, local_batch, n_tables, hot_offsets=None, offset_base=0, indices=None, offsets=None):
    from oracle import criteo_oracle
    hot = [1] * n_tables if hot_offsets is None else list(np.diff(_np(hot_offsets)))
    idx, off = criteo_oracle.regroup(_np(recv), world, local_batch, hot, offset_base)
    idx_t, off_t = torch.from_numpy(idx.astype(_np(recv).dtype)), torch.from_numpy(off.astype(_np(recv).dtype))
    if indices is not None:
        indices[:idx_t.numel()].copy_(idx_t)
        idx_t = indices
    if offsets is not None:
        offsets[:off_t.numel()].copy_(off_t)
        off_t = offsets
    return idx_t, off_t


def install(ops_module):
    for name in ("table_init_uniform_", "embbag_forward", "embbag_backward_update", "linear_forward",
                 "linear_backward_data", "linear_backward_weight", "sgd_step_", "interact_dot_forward",
                 "interact_dot_backward", "loss_mse", "loss_bce_logits", "copy_2d_", "ids_regroup"):
        setattr(ops_module, name, globals()[name])
