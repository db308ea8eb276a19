"""Table-wise sharding plan for the embedding tables of one node.

The reference is single process; the closest thing it has is the striping of
tables over parallel units (include/data2.h:150, src/emb_cache2.cpp:701-709).
Here every rank owns whole tables (table-wise model parallelism) and the MLP
is data parallel; pooled embeddings travel with one all-to-all each way.
"""
from dataclasses import dataclass
from typing import List


@dataclass
class ShardingPlan:
    world: int
    owner: List[int]                  # owner[t] = rank that holds table t
    tables_of_rank: List[List[int]]   # ascending table ids per rank
    feature_order: List[int]          # table ids in rank-major order (= feature order after the exchange)

    def local_tables(self, rank):
        return self.tables_of_rank[rank]

    def feature_slot(self, rank):
        """(first feature index, count) of the block that `rank` contributes."""
        first = sum(len(self.tables_of_rank[r]) for r in range(rank))
        return first, len(self.tables_of_rank[rank])


def plan_table_wise(table_rows, world, dim=1, bytes_per_elem=4, lookups_per_table=None, mem_cap_bytes=None):
    """Greedy bin packing, biggest table first.  A table goes to the rank with the least gather
    traffic so far (lookups), ties to the rank holding the fewest bytes, then to the lower rank;
    ranks whose HBM budget (`mem_cap_bytes`, default: no limit -- all MLPerf tables together are
    96 GB and fit one 288 GB GPU) the table would exceed are skipped.  Traffic, not footprint,
    is what costs time per step, so it is the primary load.  Deterministic."""
    T = len(table_rows)
    if world < 1:
        raise ValueError("world must be >= 1")
    lookups = list(lookups_per_table) if lookups_per_table is not None else [1.0] * T
    if len(lookups) != T:
        raise ValueError("lookups_per_table must have one entry per table")
    mem = [float(r) * dim * bytes_per_elem for r in table_rows]
    order = sorted(range(T), key=lambda t: (-lookups[t], -mem[t], t))
    load_mem = [0.0] * world
    load_lk = [0.0] * world
    owner = [0] * T
    for t in order:
        ranks = [r for r in range(world) if mem_cap_bytes is None or load_mem[r] + mem[t] <= mem_cap_bytes]
        if not ranks:
            raise ValueError("table %d (%d rows) does not fit any rank's memory budget" % (t, table_rows[t]))
        best = min(ranks, key=lambda r: (load_lk[r], load_mem[r], r))
        owner[t] = best
        load_mem[best] += mem[t]
        load_lk[best] += lookups[t]
    tables_of_rank = [[t for t in range(T) if owner[t] == r] for r in range(world)]
    feature_order = [t for r in range(world) for t in tables_of_rank[r]]
    return ShardingPlan(world, owner, tables_of_rank, feature_order)
