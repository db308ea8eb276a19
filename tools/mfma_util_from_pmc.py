#!/usr/bin/env python3
"""Matrix-pipe utilisation of the GEMM kernels of a train step from rocprofv3 PMC passes over bench.py
(summaries made by tools/prof_summary.py).

    python tools/mfma_util_from_pmc.py <mfma_b65536_pmc.json> <mfma_b8192_pmc.json> <wait_b65536_pmc.json> <out.json>

mfma passes: SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE;
wait pass: SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT
SQ_LDS_IDX_ACTIVE.  Counters are sums over the chip: MFMA busy cycles over 1024 SIMDs, GRBM_GUI_ACTIVE over 8 XCDs."""
import json
import sys


def short(k):
    k = k.replace("void dlrm::", "")
    return k[:k.index("(")] if "(" in k else k


def mfma(path):
    out = {}
    for k, c in json.load(open(path)).items():
        if "gemm_" not in k or "SQ_VALU_MFMA_BUSY_CYCLES" not in c:
            continue
        n = c["GRBM_GUI_ACTIVE"]["dispatches"]
        act = c["GRBM_GUI_ACTIVE"]["sum"] / 8.0 / n
        us = c["GRBM_GUI_ACTIVE"]["avg_ns_under_pmc"] / 1e3
        out[short(k)] = {"dispatches": n, "avg_us_under_pmc": round(us, 1),
                         "mfma_busy": round(c["SQ_VALU_MFMA_BUSY_CYCLES"]["sum"] / 1024.0 / n / act, 3),
                         "clock_GHz": round(act / (us * 1e3), 2),
                         "bf16_mfma_per_dispatch": c["SQ_INSTS_VALU_MFMA_MOPS_BF16"]["sum"] / n,
                         "f16_mfma_per_dispatch": c.get("SQ_INSTS_VALU_MFMA_MOPS_F16", {"sum": 0})["sum"] / n}
    return out


def waits(path, into):
    for k, c in json.load(open(path)).items():
        s = short(k)
        if s not in into or "SQ_WAVE_CYCLES" not in c:
            continue
        w = c["SQ_WAVE_CYCLES"]["sum"]
        into[s]["wave_states"] = {
            "parked": round(c["SQ_WAIT_ANY"]["sum"] / w, 3),
            "issue_stall": round(c["SQ_WAIT_INST_ANY"]["sum"] / w, 3),
            "of_which_lds_issue": round(c["SQ_WAIT_INST_LDS"]["sum"] / w, 3),
            "issuing": round(c["SQ_ACTIVE_INST_ANY"]["sum"] / w, 3),
            "lds_bank_conflict_over_lds_active": round(c["SQ_LDS_BANK_CONFLICT"]["sum"] / max(c["SQ_LDS_IDX_ACTIVE"]["sum"], 1), 4)}


def main():
    b64, b8, wait, out = sys.argv[1:5]
    res = {"command": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE -- "
                      "python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline [--global-batch 8192]; second pass: SQ_WAVE_CYCLES "
                      "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE "
                      "(batch 65536); tools/mfma_util_from_pmc.py",
           "note": "per GEMM kernel of the train step, averages per dispatch over all layers' launches.  mfma_busy = "
                   "(SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs) / (GRBM_GUI_ACTIVE / 8 XCDs); clock_GHz = (GRBM_GUI_ACTIVE / 8) / "
                   "kernel time under the counters (reads high on dispatches much shorter than 0.3 ms).  Wave states are "
                   "fractions of SQ_WAVE_CYCLES: parked (s_waitcnt / barrier), issue stall (matrix pipe busy or operand not "
                   "ready; of which LDS issue), issuing.",
           "batch_65536": mfma(b64), "batch_8192": mfma(b8)}
    waits(wait, res["batch_65536"])
    json.dump(res, open(out, "w"), indent=1)
    for b in ("batch_65536", "batch_8192"):
        for k, v in res[b].items():
            print(b, k, v["mfma_busy"], v["clock_GHz"], v["avg_us_under_pmc"])


if __name__ == "__main__":
    main()
