"""Sharding plans for the embedding tables of one node.

The reference is single process; the closest thing it has is the striping of
tables over parallel units (include/data2.h:150, src/emb_cache2.cpp:701-709).
Here a table is either owned whole by one rank (table-wise: pooled vectors
travel with one all-to-all each way) or split into one contiguous row range
per rank (row-wise: every rank pools the rows it holds, partial sums meet in a
reduce-scatter over the batch, gradients come back with an all-gather).  The
MLP is data parallel.
"""
from dataclasses import dataclass, field
from typing import Dict, List, Tuple


@dataclass
class ShardingPlan:
    world: int
    owner: List[int]                  # owner[t] = rank that holds table t
    tables_of_rank: List[List[int]]   # ascending table ids per rank
    feature_order: List[int]          # table ids in the order their pooled vectors sit after the exchange:
                                      # table-wise ones rank-major, then the row-wise ones
    row_wise: List[int] = field(default_factory=list)   # ascending ids of the tables split by rows
    row_ranges: Dict[int, List[Tuple[int, int]]] = field(default_factory=dict)   # [t][rank] = (first row, end row)

    def local_tables(self, rank):
        """Whole tables owned by `rank` (ascending ids)."""
        return self.tables_of_rank[rank]

    def row_range(self, table, rank):
        return self.row_ranges[table][rank]

    @property
    def n_table_wise(self):
        return len(self.feature_order) - len(self.row_wise)

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


def split_rows(rows, world):
    """Contiguous, near-equal row ranges of one table, one per rank."""
    return [(rows * r // world, rows * (r + 1) // world) for r in range(world)]


def plan_hybrid(table_rows, world, dim=1, bytes_per_elem=4, lookups_per_table=None, row_wise_min_rows=None,
                row_wise_tables=None, mem_cap_bytes=None):
    """Tables with at least `row_wise_min_rows` rows (or those listed in `row_wise_tables`) are split
    by rows over all ranks; the others are placed whole by the table-wise greedy, which starts from
    the load the row-wise shards already put on every rank."""
    T = len(table_rows)
    if row_wise_tables is None:
        row_wise_tables = [t for t in range(T) if row_wise_min_rows is not None and table_rows[t] >= row_wise_min_rows]
    rw = sorted(set(int(t) for t in row_wise_tables))
    if any(t < 0 or t >= T for t in rw):
        raise ValueError("row_wise_tables out of range")
    if any(table_rows[t] < world for t in rw):
        raise ValueError("a row-wise table needs at least one row per rank")
    tw = [t for t in range(T) if t not in rw]
    lookups = list(lookups_per_table) if lookups_per_table is not None else [1.0] * T
    cap = mem_cap_bytes
    if cap is not None:
        cap = cap - sum(float(table_rows[t]) * dim * bytes_per_elem / world for t in rw)
        if cap < 0:
            raise ValueError("row-wise shards alone exceed the memory budget")
    sub = plan_table_wise([table_rows[t] for t in tw], world, dim, bytes_per_elem, [lookups[t] for t in tw], cap)
    owner = [-1] * T
    for i, t in enumerate(tw):
        owner[t] = sub.owner[i]
    tables_of_rank = [[tw[i] for i in sub.tables_of_rank[r]] for r in range(world)]
    feature_order = [t for r in range(world) for t in tables_of_rank[r]] + rw
    return ShardingPlan(world, owner, tables_of_rank, feature_order, rw,
                        {t: split_rows(table_rows[t], world) for t in rw})
