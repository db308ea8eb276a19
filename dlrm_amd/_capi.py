"""ctypes binding of libdlrm_amd.so -- the C ABI declared in include/dlrm_amd.h.

There is no CPU fallback: if the library is missing, loading fails loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DLRM_AMD_LIB") or os.path.join(_HERE, "lib", "libdlrm_amd.so")   # (override: A/B builds)

OK = 0
F32, BF16 = 0, 1
I32, I64 = 0, 1
POOL_SUM, POOL_MEAN = 0, 1
DUP_ACCUMULATE, DUP_LAST_WINS = 0, 1
OPT_SGD, OPT_ADAGRAD_ROWWISE = 0, 1
ACT_NONE, ACT_RELU = 0, 1
GEMM_DEFAULT, GEMM_F32_MFMA, GEMM_BF16X3, GEMM_F16X2 = 0, 1, 2, 3


class DlrmError(RuntimeError):
    pass


ABI_VERSION = 1018   # DLRM_ABI_VERSION of include/dlrm_amd.h these structures mirror


class EmbBagDesc(C.Structure):
    _fields_ = [
        ("d_weight", C.c_void_p),
        ("d_table_row_offsets", C.c_void_p),
        ("total_rows", C.c_int64),
        ("num_tables", C.c_int32),
        ("dim", C.c_int32),
        ("row_dtype", C.c_int32),
        ("idx_dtype", C.c_int32),
        ("d_indices", C.c_void_p),
        ("d_offsets", C.c_void_p),
        ("n_indices", C.c_int64),
        ("batch", C.c_int64),
        ("include_last_offset", C.c_int32),
        ("pooling", C.c_int32),
        ("d_error_flag", C.c_void_p),
        ("d_shard_row_begin", C.c_void_p),
    ]


class SparseOpt(C.Structure):
    _fields_ = [
        ("optimizer", C.c_int32),
        ("dup_mode", C.c_int32),
        ("lr", C.c_float),
        ("eps", C.c_float),
        ("d_opt_state", C.c_void_p),
    ]


class LinearDesc(C.Structure):
    _fields_ = [
        ("M", C.c_int64), ("N", C.c_int32), ("K", C.c_int32),
        ("d_x", C.c_void_p), ("x_ld", C.c_int64),
        ("d_w", C.c_void_p), ("w_ld", C.c_int64),
        ("d_bias", C.c_void_p),
        ("d_y", C.c_void_p), ("y_ld", C.c_int64),
        ("act", C.c_int32),
        ("gemm_mode", C.c_int32),
        ("d_x_absmax", C.c_void_p), ("d_w_absmax", C.c_void_p), ("d_y_absmax", C.c_void_p),
        ("w_planes", C.c_void_p),
    ]


class LinearBwdDesc(C.Structure):
    _fields_ = [
        ("M", C.c_int64), ("N", C.c_int32), ("K", C.c_int32),
        ("d_dz", C.c_void_p), ("dz_ld", C.c_int64),
        ("d_w", C.c_void_p), ("w_ld", C.c_int64),
        ("d_x", C.c_void_p), ("x_ld", C.c_int64),
        ("d_dx", C.c_void_p), ("dx_ld", C.c_int64),
        ("d_relu_mask", C.c_void_p), ("relu_mask_ld", C.c_int64),
        ("d_dw", C.c_void_p), ("dw_ld", C.c_int64),
        ("d_db", C.c_void_p),
        ("gemm_mode", C.c_int32),
        ("d_dz_absmax", C.c_void_p), ("d_w_absmax", C.c_void_p), ("d_x_absmax", C.c_void_p), ("d_dx_absmax", C.c_void_p),
        ("w_planes", C.c_void_p),
    ]


class WeightPlanes(C.Structure):
    _fields_ = [
        ("d_w", C.c_void_p), ("w_ld", C.c_int64), ("N", C.c_int32), ("K", C.c_int32),
        ("d_w_absmax", C.c_void_p), ("d_planes", C.c_void_p),
    ]


class FeatureView(C.Structure):
    _fields_ = [("d_base", C.c_void_p), ("stride", C.c_int64)]


class CopyItem(C.Structure):
    _fields_ = [
        ("d_dst", C.c_void_p), ("dst_ld", C.c_int64),
        ("d_src", C.c_void_p), ("src_ld", C.c_int64),
        ("rows", C.c_int64), ("cols", C.c_int64),
    ]


_lib = None


def lib():
    """Loads libdlrm_amd.so (built in-tree by `make -C dlrm_amd/csrc`)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DlrmError(
                "libdlrm_amd.so not found at %s: build it with `make -C dlrm_amd/csrc` "
                "(or `python -c 'import __graft_entry__ as g; g.build()'`). "
                "There is no CPU fallback." % LIB_PATH)
        # torch ships its own HIP runtime (same soname as /opt/rocm's).  Import it
        # first so that libdlrm_amd.so binds to that one runtime: two runtimes in
        # one process cannot share streams or device memory.
        import torch  # noqa: F401
        _lib = C.CDLL(LIB_PATH)
        _lib.dlrm_last_error.restype = C.c_char_p
        _lib.dlrm_target_arch.restype = C.c_char_p
        # the ctypes structures below mirror one revision of the header
        if int(_lib.dlrm_version()) != ABI_VERSION:
            v = int(_lib.dlrm_version())
            _lib = None
            raise DlrmError("libdlrm_amd.so has ABI version %d, this binding expects %d: rebuild the library "
                            "(`make -C dlrm_amd/csrc`)" % (v, ABI_VERSION))
    return _lib


def check(rc):
    if rc != OK:
        raise DlrmError("dlrm_amd error %d: %s" % (rc, lib().dlrm_last_error().decode()))


def declared_symbols():
    """Every function include/dlrm_amd.h declares (parsed from the header)."""
    import re
    hdr = open(os.path.join(os.path.dirname(_HERE), "include", "dlrm_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(dlrm_[a-z0-9_]+)\s*\(", hdr)))
