"""Backing-store image in the reference's layout (dlrm_amd/store_file.py): byte-level checks."""
import numpy as np

from dlrm_amd import store_file as sf


def test_image_layout_and_padding(tmp_path):
    rng = np.random.default_rng(0)
    tables = [rng.random((100, 32), dtype=np.float32), rng.random((7, 32), dtype=np.float32)]
    p = tmp_path / "img.bin"
    n = sf.write_store(p, tables)
    raw = p.read_bytes()
    assert n == len(raw) == sf.image_bytes([100, 7], 32) == 32768 and n % (8 * 4096) == 0
    # row-major little-endian float32, tables back to back
    assert raw[:4] == tables[0][0, :1].astype("<f4").tobytes()
    assert np.frombuffer(raw[100 * 32 * 4:100 * 32 * 4 + 7 * 32 * 4], "<f4").tolist() == tables[1].reshape(-1).tolist()
    assert set(raw[107 * 32 * 4:]) == {ord("0")}              # padded with the character '0'
    # one cache page (4096 bytes) is rows 32k .. 32k+31 of a dim-32 table
    assert np.frombuffer(raw[4096:8192], "<f4").reshape(32, 32).tolist() == tables[0][32:64].tolist()
    back = sf.read_store(p, [100, 7], 32, copy=True)
    assert all(np.array_equal(a, b) for a, b in zip(back, tables))


def test_exact_multiple_needs_no_padding(tmp_path):
    t = np.arange(256 * 32, dtype=np.float32).reshape(256, 32)       # 32 KiB
    p = tmp_path / "img.bin"
    assert sf.write_store(p, [t]) == 32768
    assert np.array_equal(sf.read_store(p, [256], 32)[0], t)
