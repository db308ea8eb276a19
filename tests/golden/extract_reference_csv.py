"""Turns the run artefacts the reference ships in dlrm_libtorch/build/ into a
small fixture: per batch, the total pages read and pages written over all CSV
rows (the reference prints them per physical PU; physical placement is out of
scope, totals are placement independent).

Run in the build container (needs /root/reference):
    python tests/golden/extract_reference_csv.py
"""
import json
import os

REF = "/root/reference/dlrm_libtorch/build"
# name -> (csv, T, D, L); all runs: batch 40, data_size 200, cache mult 2, 2 epochs,
# 500000 rows per table (the RMC presets in src/dlrmtest.cpp:15-36)
RUNS = {
    "rmc1": ("LRU_BASE_RMC1_C2_B40_D200.csv", 32, 32, 80),
    "rmc2": ("LRU_BASE_RMC2_C2_B40_D200.csv", 32, 64, 120),
    "rmc3": ("LRU_BASE_RMC3_C2_B40_D200.csv", 40, 32, 20),
}


# dlrmtest2 (class cache2) runs.  The file names carry the preset label of the day; the table count is
# the number of rows per block and the other sizes are the presets of src/dlrmtest2.cpp:19-46 that
# reproduce the first batch's cold reads.  LRU_RMC3 is byte-for-byte the counters of LRU_RMC1.
VEC_RUNS = {
    "vec_t32_d64_l120": ("LRU_RMC1_C2_B40_D200.csv", 32, 64, 120, 500000),
    "vec_t25_d32_l20": ("LRU_RMC2_C2_B40_D200.csv", 25, 32, 20, 800000),
}


def vec_main():
    """Per batch and per table (the CSV's channel/PU columns are table / 8 and table % 8,
    src/dlrmtest2.cpp:228, :476): pages read by emb::emb and pages written by reclaim_emb_cache."""
    out = {}
    for name, (csv, T, D, L, R) in VEC_RUNS.items():
        rows = [r.strip().split(";") for r in open(os.path.join(REF, csv))][1:]
        batches = []
        i = 0
        while i < len(rows):
            rd = rows[i:i + T]
            wr = rows[i + T:i + 2 * T]
            assert all(len(r) == 4 for r in rd) and all(len(r) == 5 for r in wr)
            assert [int(r[1]) * 8 + int(r[2]) for r in rd] == list(range(T))
            batches.append({"read": [int(r[3]) for r in rd], "write": [int(r[4]) for r in wr]})
            i += 2 * T
        out[name] = {"source": "dlrm_libtorch/build/" + csv, "tables": T, "dim": D, "lookups": L,
                     "batch": 40, "data_size": 200, "cache_mult": 2, "epochs": 2,
                     "rows_per_table": R, "batches": batches}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_lru_vec_trace.json")
    json.dump(out, open(path, "w"))
    print("wrote", path, {k: len(v["batches"]) for k, v in out.items()})


def main():
    out = {}
    for name, (csv, T, D, L) in RUNS.items():
        rows = [r.strip().split(";") for r in open(os.path.join(REF, csv))][1:]
        batches = []
        i = 0
        while i < len(rows):
            rd = rows[i:i + T]
            wr = rows[i + T:i + 2 * T]
            assert all(len(r) == 4 for r in rd) and all(len(r) == 5 for r in wr)
            batches.append({"read": sum(int(r[3]) for r in rd), "write": sum(int(r[4]) for r in wr)})
            i += 2 * T
        out[name] = {"source": "dlrm_libtorch/build/" + csv, "tables": T, "dim": D, "lookups": L,
                     "batch": 40, "data_size": 200, "cache_mult": 2, "epochs": 2,
                     "rows_per_table": 500000, "batches": batches}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_lru_base_totals.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path, {k: len(v["batches"]) for k, v in out.items()})


if __name__ == "__main__":
    main()
    vec_main()
