"""The reference's backing-store image (what embedding_table_init::concat_emb assembles before it is
written to the device, src/emb_init.cpp:17-70 with src/data_transfer.cpp:17-61): every table row-major,
float32 little-endian, tables back to back, the total padded with ASCII '0' bytes to a whole write unit
(ws_opt sectors of 4096 bytes = 32 KiB).  One page of the cache (4096 bytes = 4096/(4*dim) rows) is one
sector of this image.

    write_store(path, tables)                 tables: list of [rows, dim] float32 arrays
    read_store(path, table_rows, dim)  ->     list of arrays (views of one np.memmap unless copy=True)
    load_into(cache, path) / save_from(cache, path)   move an image into / out of an EmbeddingCache's
                                                       pinned host store (flushes the cache first)
"""
import numpy as np

SECTOR_BYTES = 4096
WS_OPT = 8                                   # sectors per write unit on the reference's device
WRITE_UNIT = SECTOR_BYTES * WS_OPT


def image_bytes(table_rows, dim):
    n = sum(int(r) for r in table_rows) * dim * 4
    return (n + WRITE_UNIT - 1) // WRITE_UNIT * WRITE_UNIT


def write_store(path, tables):
    dim = tables[0].shape[1]
    total = 0
    with open(path, "wb") as f:
        for t in tables:
            a = np.ascontiguousarray(t, dtype="<f4")
            if a.ndim != 2 or a.shape[1] != dim:
                raise ValueError("every table must be [rows, %d]" % dim)
            f.write(a.tobytes())
            total += a.nbytes
        pad = image_bytes([t.shape[0] for t in tables], dim) - total
        f.write(b"0" * pad)                  # the reference pads with the character '0'
    return total + pad


def read_store(path, table_rows, dim, copy=False):
    need = image_bytes(table_rows, dim)
    mm = np.memmap(path, dtype=np.uint8, mode="r")
    if mm.size < sum(int(r) for r in table_rows) * dim * 4:
        raise ValueError("%s holds %d bytes, %d tables of dim %d need %d" % (path, mm.size, len(table_rows), dim, need))
    out, o = [], 0
    for r in table_rows:
        n = int(r) * dim * 4
        a = mm[o:o + n].view("<f4").reshape(int(r), dim)
        out.append(np.array(a) if copy else a)
        o += n
    return out


def load_into(cache, path):
    """Fill the cache's pinned store from an image; nothing may be resident (fresh cache)."""
    if cache.pool.dtype != __import__("torch").float32:
        raise ValueError("the image format is float32")
    if int(cache.counters()["resident"].sum()) != 0:
        raise ValueError("load the image before the first lookup")
    for dst, src in zip(cache.store, read_store(path, cache.table_rows, cache.dim)):
        np.copyto(dst, src)


def save_from(cache, path):
    """Write resident pages back, then dump the store as an image."""
    cache.flush()
    return write_store(path, [np.asarray(s) for s in cache.store])
