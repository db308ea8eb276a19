"""tools/cache_trace.py on a hand-written log in the reference's CSV schema."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load():
    spec = importlib.util.spec_from_file_location("cache_trace", os.path.join(ROOT, "tools", "cache_trace.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_pages_groups_read_and_write_rows_into_batches(tmp_path):
    p = tmp_path / "log.csv"
    p.write_text("'Time';'Channel';'PU';'Read';'Write'\n"
                 "0.5;0;0;10\n0.5;0;1;7\n"            # batch 0: reads of two tables
                 "0.6;0;0;;8\n0.6;0;1;;8\n"            # its reclaim
                 "0.9;0;0;3\n0.9;0;1;0\n"              # batch 1 (no reclaim rows)
                 "1.2;0;0;1\n1.2;0;1;2\n1.3;0;0;;16\n1.3;0;1;;0\n")
    b = load().read_pages(str(p), tables=2)
    assert [(x["read"], x["write"]) for x in b] == [(17, 16), (3, 0), (3, 16)]


def test_trace_hit_rates(tmp_path):
    p = tmp_path / "t.csv"
    p.write_text("'Time';'Epoch';'Batch';'Table';'Hit';'Miss';'Read';'Write';'Evicted';'Resident'\n"
                 "0.1;1;0;0;0;40;30;0;0;30\n0.1;1;0;1;0;40;35;0;0;35\n"
                 "0.2;1;1;0;30;10;8;8;8;30\n0.2;1;1;1;20;20;16;16;16;35\n")
    t = load().read_trace(str(p))
    assert [x["read_hit_rate"] for x in t] == [0.0, 50 / 80]
    assert [x["read"] for x in t] == [65, 24] and [x["evicted"] for x in t] == [0, 24]
