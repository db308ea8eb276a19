"""The reference-shaped C++ driver (adapter/dlrmtest_amd.cpp over the C ABI) run as a
process on the GPU:
  * a small model, as shipped: per-table page-traffic CSV bit-exact against the oracle's cache
    state machine, loss trace and the trained table 0 within 1e-5 of the oracle's loop;
  * the reference's RMC1 run: per-batch page totals equal the reference's own CSV."""
import json
import os
import struct
import subprocess

import numpy as np
import pytest

from tests.oracle_model import as_shipped_step
from tests.refdata import reference_batches, reference_dataset

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "adapter", "bin", "dlrmtest_amd")
GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_lru_base_totals.json")))


def write_input(path, data, target, W, b, T, D, rows, L, dense):
    with open(path, "wb") as f:
        f.write(struct.pack("<8i", 0x444c524d, data.shape[0], T, D, rows, L, dense, 0))
        f.write(np.ascontiguousarray(data, dtype=np.float32).tobytes())
        f.write(np.ascontiguousarray(target, dtype=np.float32).tobytes())
        for w, bb in zip(W, b):
            f.write(np.ascontiguousarray(w, dtype=np.float32).tobytes())
            f.write(np.ascontiguousarray(bb, dtype=np.float32).tobytes())


def run_driver(args, cwd):
    assert os.path.exists(BIN), "build the adapter first (python -c 'import __graft_entry__ as g; g.build()')"
    r = subprocess.run([BIN] + [str(a) for a in args], cwd=cwd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    return [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]


def parse_csv(path, T):
    rows = [r.strip().split(";") for r in open(path)][1:]
    out, i = [], 0
    while i < len(rows):
        out.append(([int(r[3]) for r in rows[i:i + T]], [int(r[4]) for r in rows[i + T:i + 2 * T]]))
        i += 2 * T
    return out


def test_driver_as_shipped_small(oracle, tmp_path):
    T, D, rows, L, dense, N, B, c, epochs = 4, 32, 3000, 6, 16, 50, 10, 2, 2
    rng = np.random.default_rng(1)
    ids = rng.integers(0, rows, size=(N, T * L))
    x = rng.standard_normal((N, dense)).astype(np.float32)
    target = rng.standard_normal(N).astype(np.float32)
    cat = 32 + T * D
    W = [(rng.standard_normal((128, dense)) / 4).astype(np.float32), (rng.standard_normal((32, 128)) / 11).astype(np.float32),
         (rng.standard_normal((32, cat)) / 12).astype(np.float32), (rng.standard_normal((1, 32)) / 5).astype(np.float32)]
    b = [np.abs(rng.standard_normal(n)).astype(np.float32) * 0.1 for n in (128, 32, 32, 1)]
    data = np.concatenate([x, ids.astype(np.float32)], axis=1)
    inp, csv, dump = tmp_path / "in.bin", tmp_path / "out.csv", tmp_path / "t0.bin"
    write_input(inp, data, target, W, b, T, D, rows, L, dense)
    lines = run_driver([N, B, c, "--tables", T, "--dim", D, "--rows", rows, "--lookups", L, "--dense", dense,
                        "--epochs", epochs, "--input", inp, "--csv", csv, "--dump-table0", dump], tmp_path)

    # the same loop on the oracle
    tables = [oracle.table_init_uniform(rows * D, 7919 * (t + 1)).reshape(rows, D) for t in range(T)]
    pc = oracle.PageCache(T, rows, D, cache_size=c * T * L * B)
    ref_csv, ref_loss = [], []
    for epoch in range(1, epochs + 1):
        for bi, (s, bs) in enumerate(reference_batches(N, B)):
            for t in range(T):
                pc.get(t, np.sort(ids[s:s + bs, t * L:(t + 1) * L].reshape(-1)))
            reads = pc.counters("read", reset=True)
            pc.reclaim()
            writes = pc.counters("write", reset=True)
            if bi + 1 == c and epoch == 1:
                pc.cache_size = pc.page_count
            ref_csv.append((reads.tolist(), writes.tolist()))
            loss, _ = as_shipped_step(oracle, tables, ids[s:s + bs], x[s:s + bs], target[s:s + bs], W, b, L)
            ref_loss.append(loss)
    assert parse_csv(csv, T) == ref_csv
    got_loss = [l["loss"] for l in lines if "loss" in l]
    assert len(got_loss) == len(ref_loss)
    assert np.allclose(got_loss, ref_loss, rtol=1e-5, atol=1e-7)
    assert lines[-1]["error_flags"] == 0
    t0 = np.fromfile(dump, dtype=np.float32).reshape(rows, D)
    assert np.max(np.abs(t0 - tables[0])) <= 1e-5 * np.max(np.abs(tables[0]))
    assert not np.array_equal(t0, oracle.table_init_uniform(rows * D, 7919).reshape(rows, D))   # it did train


def test_driver_rmc1_matches_reference_csv(tmp_path):
    cfg = GOLD["rmc1"]
    T, D, L = cfg["tables"], cfg["dim"], cfg["lookups"]
    B, N, c, rows = cfg["batch"], cfg["data_size"], cfg["cache_mult"], cfg["rows_per_table"]
    x, ids, y = reference_dataset(T, L, B, N, rows)
    rng = np.random.default_rng(0)
    cat = 32 + T * D
    W = [(rng.standard_normal((128, 256)) / 16).astype(np.float32), (rng.standard_normal((32, 128)) / 11).astype(np.float32),
         (rng.standard_normal((32, cat)) / 32).astype(np.float32), (rng.standard_normal((1, 32)) / 5).astype(np.float32)]
    b = [np.abs(rng.standard_normal(n)).astype(np.float32) * 0.1 for n in (128, 32, 32, 1)]
    inp, csv = tmp_path / "in.bin", tmp_path / "out.csv"
    write_input(inp, np.concatenate([x, ids.astype(np.float32)], axis=1), y.reshape(-1), W, b, T, D, rows, L, 256)
    trace = tmp_path / "out.trace.csv"
    lines = run_driver([N, B, c, "--input", inp, "--csv", csv, "--trace", trace], tmp_path)
    got = [{"read": sum(r), "write": sum(w)} for r, w in parse_csv(csv, T)]
    assert got == cfg["batches"]
    # the log tooling sees the same totals in both files, and every lookup is a hit or a miss
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location(
        "cache_trace", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "cache_trace.py"))
    ct = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ct)
    pages, tr = ct.read_pages(str(csv), T), ct.read_trace(str(trace))
    assert [{"read": p["read"], "write": p["write"]} for p in pages] == cfg["batches"]
    assert [t["read"] for t in tr] == [p["read"] for p in pages]
    assert [t["write"] for t in tr] == [p["write"] for p in pages]
    sizes = [min(B, N - s) for s in range(0, N, B)]
    assert [t["hit"] + t["miss"] for t in tr] == [T * L * n for _ in range(len(tr) // len(sizes)) for n in sizes]
    assert lines[-1]["error_flags"] == 0
    assert all(np.isfinite(l["loss"]) for l in lines if "loss" in l)


VEC_GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_lru_vec_trace.json")))


def test_driver_vector_policy_matches_reference_dlrmtest2_csv(tmp_path):
    """--policy vec on the reference's dlrmtest2 run (25 tables x 800000 rows x 32, 20 lookups): the CSV the
    driver writes carries, per batch and table, the page reads and writes of the reference's own CSV."""
    cfg = VEC_GOLD["vec_t25_d32_l20"]
    T, D, L = cfg["tables"], cfg["dim"], cfg["lookups"]
    B, N, c, rows = cfg["batch"], cfg["data_size"], cfg["cache_mult"], cfg["rows_per_table"]
    x, ids, y = reference_dataset(T, L, B, N, rows)
    rng = np.random.default_rng(0)
    cat = 32 + T * D
    W = [(rng.standard_normal((128, 256)) / 16).astype(np.float32), (rng.standard_normal((32, 128)) / 11).astype(np.float32),
         (rng.standard_normal((32, cat)) / 32).astype(np.float32), (rng.standard_normal((1, 32)) / 5).astype(np.float32)]
    b = [np.abs(rng.standard_normal(n)).astype(np.float32) * 0.1 for n in (128, 32, 32, 1)]
    inp, csv = tmp_path / "in.bin", tmp_path / "out.csv"
    write_input(inp, np.concatenate([x, ids.astype(np.float32)], axis=1), y.reshape(-1), W, b, T, D, rows, L, 256)
    lines = run_driver([N, B, c, "--policy", "vec", "--tables", T, "--dim", D, "--rows", rows, "--lookups", L,
                        "--input", inp, "--csv", csv, "--slots", 4_500_000, "--log-pages", 512], tmp_path)
    got = [{"read": r, "write": w} for r, w in parse_csv(csv, T)]
    assert got == cfg["batches"]
    assert lines[-1]["error_flags"] == 0
    assert all(np.isfinite(l["loss"]) for l in lines if "loss" in l)


def test_driver_vector_policy_small(oracle, tmp_path):
    """Small model, --canonical (every table trained) under --policy vec with a list short enough to evict:
    the CSV equals the oracle's vector cache fed with the same calls, rows that were never asked for keep
    their initial value wherever the cache has moved them, the asked ones train."""
    T, D, rows, L, dense, N, B, c, epochs = 3, 32, 4000, 40, 16, 60, 10, 1, 2
    rng = np.random.default_rng(2)
    ids = rng.integers(0, rows, size=(N, T * L))
    x = rng.standard_normal((N, dense)).astype(np.float32)
    target = rng.standard_normal(N).astype(np.float32)
    cat = 32 + T * D
    W = [(rng.standard_normal((128, dense)) / 4).astype(np.float32), (rng.standard_normal((32, 128)) / 11).astype(np.float32),
         (rng.standard_normal((32, cat)) / 12).astype(np.float32), (rng.standard_normal((1, 32)) / 5).astype(np.float32)]
    b = [np.abs(rng.standard_normal(n)).astype(np.float32) * 0.1 for n in (128, 32, 32, 1)]
    inp = tmp_path / "in.bin"
    write_input(inp, np.concatenate([x, ids.astype(np.float32)], axis=1), target, W, b, T, D, rows, L, dense)
    common = [N, B, c, "--tables", T, "--dim", D, "--rows", rows, "--lookups", L, "--dense", dense, "--epochs", epochs,
              "--input", inp, "--canonical"]
    vec = run_driver(common + ["--policy", "vec", "--csv", tmp_path / "v.csv", "--dump-table0", tmp_path / "v0.bin"], tmp_path)
    assert vec[-1]["error_flags"] == 0
    lv = [l["loss"] for l in vec if "loss" in l]
    assert len(lv) == epochs * 6
    v0 = np.fromfile(tmp_path / "v0.bin", dtype=np.float32)
    init = oracle.table_init_uniform(rows * D, 7919)
    assert v0.shape == init.shape
    assert not np.array_equal(v0, init) and np.all(np.isfinite(v0)) and np.all(np.isfinite(lv))
    # rows the run never asked for keep their initial value, wherever the cache has moved them
    asked = np.zeros(rows, bool)
    asked[ids[:, :L].reshape(-1)] = True
    assert np.array_equal(v0.reshape(rows, D)[~asked], init.reshape(rows, D)[~asked])
    assert np.any(v0.reshape(rows, D)[asked] != init.reshape(rows, D)[asked])
    # cache traffic against the oracle's vector cache (c * L * B = 400 listed rows per table: evictions happen)
    vc = oracle.VectorCache(T, rows, D, c * L * B)
    ref = []
    for epoch in range(epochs):
        for s, bs in reference_batches(N, B):
            req = [ids[s:s + bs, t * L:(t + 1) * L].reshape(-1) for t in range(T)]
            for t in range(T):
                vc.get(t, req[t])
            reads = vc.counters("read", reset=True)
            vc.reclaim()
            writes = vc.counters("write", reset=True)
            for t in range(T):
                vc.save(t, req[t])
            ref.append((reads.tolist(), writes.tolist()))
    assert parse_csv(tmp_path / "v.csv", T) == ref
    assert sum(sum(w) for _, w in ref) > 0


def test_cpp_train_host_single_and_forced_exchange():
    """adapter/bin/dlrm_train: the Criteo-shaped step hosted in C++ over dlrm_model_* (no Python in the process).
    One rank without exchanges, the same with every exchange of the sharded schedule over a dlrm_comm (RCCL)
    communicator of one rank, and with the large tables split by rows: same batches, same seeds, so the loss after
    the last step of the first two must agree to float32 rounding."""
    exe = os.path.join(ROOT, "adapter", "bin", "dlrm_train")
    base = [exe, "--config", "kaggle", "--row-cap", "20000", "--batch", "1024", "--steps", "6", "--warmup", "2"]
    losses = []
    for extra in ([], ["--force-exchange"], ["--force-exchange", "--row-wise-min-rows", "20000"]):
        out = subprocess.run(base + extra, capture_output=True, text=True, timeout=180)
        assert out.returncode == 0, out.stderr
        rec = json.loads(out.stdout.strip().splitlines()[-1])
        assert rec["samples_per_s"] > 0 and np.isfinite(rec["loss_share"]) and 0.0 < rec["loss_share"] < 5.0
        assert rec["exchange"] == bool(extra)
        losses.append(rec["loss_share"])
    # (with row-wise tables the interaction sees the features in another order: another model from the same seed)
    assert abs(losses[0] - losses[1]) <= 2e-5 * abs(losses[0]), losses
