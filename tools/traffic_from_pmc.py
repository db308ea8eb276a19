#!/usr/bin/env python3
"""Aggregate two rocprofv3 PMC passes over `bench.py` (one with FETCH_SIZE, one with WRITE_SIZE)
into HBM bytes per operator launch.

    python tools/traffic_from_pmc.py <fetch> <write> <steps_run> <out.json> [workload]

<fetch> / <write>: a rocprofv3 output directory, or the per-kernel summary tools/prof_summary.py made of one
(*_pmc.json).  [workload] (e.g. "mlperf:65536") is stored with the numbers; bench.py attaches them only to a run of
that workload.

FETCH_SIZE / WRITE_SIZE are kilobytes at the L2's memory side.  On gfx950 FETCH_SIZE tallies the
128-byte requests of wide coalesced reads at 64 bytes, so it is doubled.  `steps_run` is
warmup + steps of the profiled command when every launch belongs to a train step (0 = the run also
launched operators outside train steps, e.g. bench.py's inference batches: only per-launch averages are kept)."""
import collections
import csv
import glob
import json
import sys

# kernel-name fragment -> (operator, counts as one operator launch?)
RULES = [
    ("gemm_f16q_kernel<true, false>", "linear_forward", True),           # (TAIL, W transposed)
    ("gemm_f16q_kernel<false, false>", "linear_forward", True),
    ("gemm_f16q_kernel<true, true>", "linear_backward_data", True),
    ("gemm_f16q_kernel<false, true>", "linear_backward_data", True),
    ("gemm_f16p_kernel<true, true, true, 4, 0, 2, false>", "linear_forward", True),     # (… TAIL, paths, NI, -, -, W transposed)
    ("gemm_f16p_kernel<false, true, true, 4, 0, 2, false>", "linear_forward", True),
    ("gemm_f16p_kernel<true, true, true, 4, 0, 2, true>", "linear_backward_data", True),
    ("gemm_f16p_kernel<false, true, true, 4, 0, 2, true>", "linear_backward_data", True),
    ("gemm_split_kernel<true, true", "linear_forward", True),
    ("gemm_f32_kernel<true, true", "linear_forward", True),
    ("gemm_split_kernel<true, false", "linear_backward_data", True),
    ("gemm_f32_kernel<true, false", "linear_backward_data", True),
    ("gemm_split_kernel<false, false", "linear_backward_weight", True),
    ("gemm_f32_kernel<false, false", "linear_backward_weight", True),
    ("reduce_slabs_kernel", "linear_backward_weight", False),
    ("absmax_kernel", "operand_maxima", True),
    ("absmax_many_kernel", "operand_maxima", True),
    ("split_weights_kernel", "operand_maxima", False),
    ("skinny_k_fwd_kernel", "linear_forward_skinny", True),
    ("gemv_fwd_kernel", "linear_forward_skinny", True),
    ("gemv_fwd_loss_bwd_kernel", "linear_forward_skinny", True),
    ("gemv_bwd_data_kernel", "linear_backward_data_skinny", True),
    ("skinny_k_bwd_weight_kernel", "linear_backward_weight_skinny", True),
    ("gemv_bwd_weight_kernel", "linear_backward_weight_skinny", True),
    ("embbag_fwd", "embbag_forward", True),
    ("expand_pairs_kernel", "embbag_backward_prepare", True),    # ids only: pairs + sort, beside the forward pass
    ("rocprim", "embbag_backward_prepare", False),
    ("digit_hist_kernel", "embbag_backward_prepare", False),
    ("digit_offsets_kernel", "embbag_backward_prepare", False),
    ("digit_scatter_kernel", "embbag_backward_prepare", False),
    ("sparse_update_kernel", "embbag_backward_update", True),
    ("combine_runs_kernel", "embbag_backward_update", False),
    ("interact_dot_fwd", "interact_dot_forward", True),
    ("interact_dot_bwd", "interact_dot_backward", True),
    ("sgd_step_kernel", "sgd_step", True),
    ("loss_", "loss", True),
]


def op_of(kernel):
    for frag, op, anchor in RULES:
        if frag in kernel:
            return op, anchor
    return None, False


def read(src, counter):
    tot = collections.defaultdict(float)
    launches = collections.defaultdict(int)
    per_kernel = collections.defaultdict(lambda: [0.0, 0])

    def add(k, kib, n):
        v = kib * 1024.0
        per_kernel[k][0] += v
        per_kernel[k][1] += n
        op, anchor = op_of(k)
        if op is None:
            return
        tot[op] += v
        if anchor:
            launches[op] += n

    if src.endswith(".json"):
        for k, ctrs in json.load(open(src)).items():
            if counter in ctrs:
                add(k, ctrs[counter]["sum"], ctrs[counter]["dispatches"])
    else:
        for f in glob.glob(src + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == counter:
                    add(r["Kernel_Name"], float(r["Counter_Value"]), 1)
    return tot, launches, per_kernel


def main():
    fetch_dir, write_dir, steps, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    rd, launches, pk_r = read(fetch_dir, "FETCH_SIZE")
    wr, _, pk_w = read(write_dir, "WRITE_SIZE")
    res = {"note": "HBM-side bytes per operator launch; reads = FETCH_SIZE x 1024 x 2 (gfx950 tallies 128-byte "
                   "read requests at 64 bytes), writes = WRITE_SIZE x 1024; separate rocprofv3 --pmc passes",
           "workload": sys.argv[5] if len(sys.argv) > 5 else None,
           "steps_profiled": steps, "per_op_bytes_per_launch": {}, "per_op": {}, "per_kernel": {}}
    for op in sorted(set(rd) | set(wr)):
        n = max(launches.get(op, 0), 1)
        r, w = 2.0 * rd.get(op, 0.0) / n, wr.get(op, 0.0) / n
        res["per_op_bytes_per_launch"][op] = r + w
        res["per_op"][op] = {"launches": n, "read_bytes_per_launch": r, "write_bytes_per_launch": w}
        if steps > 0:
            res["per_op"][op]["launches_per_step"] = n / steps
    for k in sorted(set(pk_r) | set(pk_w)):
        r, n = pk_r.get(k, [0.0, 0])
        w, n2 = pk_w.get(k, [0.0, 0])
        n = max(n, n2, 1)
        res["per_kernel"][k[:120]] = {"dispatches": n, "read_bytes_per_dispatch": 2.0 * r / n,
                                      "write_bytes_per_dispatch": w / n}
    json.dump(res, open(out, "w"), indent=1)
    for op, b in res["per_op_bytes_per_launch"].items():
        print("%-28s %10.1f MB/launch" % (op, b / 1e6))


if __name__ == "__main__":
    main()
