"""ctypes front-end to the CHECKER libraries.  TEST INFRASTRUCTURE ONLY.

* ``Oracle``    -- oracle/liboracle.so, the CPU restatement (smatrix_oracle.c)
* ``Reference`` -- oracle/_ref/libsmatrix_ref.so, the real reference compiled
  from its own sources by oracle/Makefile (+ ref_probe.c introspection shim)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  Both classes expose the same methods so that differential tests
can drive them interchangeably.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "liboracle.so")
REF_SO = os.path.join(HERE, "_ref", "libsmatrix_ref.so")

OP_GET, OP_SET, OP_INCR, OP_DECR = 0, 1, 2, 3

_u32p = C.POINTER(C.c_uint32)


def build(quiet=True):
    """make -C oracle (liboracle.so always; _ref only when /root/reference exists)."""
    subprocess.run(["make", "-C", HERE], check=True,
                   stdout=subprocess.DEVNULL if quiet else None,
                   stderr=subprocess.DEVNULL if quiet else None)


def have_reference():
    return os.path.exists(REF_SO)


def _u32(a):
    return np.ascontiguousarray(a, dtype=np.uint32)


def _ptr(a):
    return a.ctypes.data_as(_u32p)


class _Base:
    """Shared driver; subclasses bind names (prefix differs: ora_ / smatrix_+ref_)."""

    _lib = None

    def __init__(self, fname=None):
        self._h = self._open(fname.encode() if fname else None)
        if not self._h:
            raise OSError("open failed: %r" % (fname,))

    # scalar API -----------------------------------------------------------
    def get(self, x, y):
        return self._get(self._h, x, y)

    def set(self, x, y, v):
        return self._set(self._h, x, y, v)

    def incr(self, x, y, v):
        return self._incr(self._h, x, y, v)

    def decr(self, x, y, v):
        return self._decr(self._h, x, y, v)

    def rowlen(self, x):
        return self._rowlen(self._h, x)

    def getrow(self, x, ret_len_bytes=None):
        """Returns the (n,2) uint32 pairs exactly as written into the buffer."""
        if ret_len_bytes is None:
            ret_len_bytes = max(self.rowlen(x) + 1, 1) * 8
        buf = np.zeros(ret_len_bytes // 4 + 4, dtype=np.uint32)  # +slack: S4 overrun
        n = self._getrow(self._h, x, _ptr(buf), ret_len_bytes)
        return buf[: 2 * n].reshape(n, 2).copy()

    def close(self):
        if self._h:
            self._close(self._h)
            self._h = None

    # batch drivers --------------------------------------------------------
    def apply(self, op, x, y, v=None):
        x, y = _u32(x), _u32(y)
        v = _u32(v) if v is not None else np.zeros_like(x)
        out = np.empty_like(x)
        self._apply(self._h, op, x.size, _ptr(x), _ptr(y), _ptr(v), _ptr(out))
        return out

    def sum_get(self, x, y):
        x, y = _u32(x), _u32(y)
        return int(self._sum_get(self._h, x.size, _ptr(x), _ptr(y)))

    # introspection --------------------------------------------------------
    def num_rows(self):
        return int(self._num_rows(self._h))

    def dir_size(self):
        return int(self._dir_size(self._h))

    def mem(self):
        return int(self._mem(self._h))

    def row_info(self, x):
        size, used = C.c_uint32(0), C.c_uint32(0)
        ok = self._row_info(self._h, x, C.byref(size), C.byref(used))
        return (size.value, used.value) if ok else None

    def row_slots(self, x):
        info = self.row_info(x)
        if info is None:
            return None
        kv = np.zeros(2 * info[0], dtype=np.uint32)
        self._row_slots(self._h, x, _ptr(kv), info[0])
        return kv.reshape(-1, 2)

    def list_rows(self):
        n = self.num_rows()
        xs = np.zeros(max(n, 1), dtype=np.uint32)
        got = self._list_rows(self._h, _ptr(xs), n)
        return xs[:got]

    def dump(self):
        """{x: (size, used, sorted non-empty (key,value) pairs)} for every row."""
        out = {}
        for x in self.list_rows().tolist():
            size, used = self.row_info(x)
            kv = self.row_slots(x)
            ne = kv[(kv[:, 0] != 0) | (kv[:, 1] != 0)]
            out[x] = (size, used, sorted(map(tuple, ne.tolist())))
        return out


def _bind(lib, names):
    sig = {
        "open": (C.c_void_p, [C.c_char_p]),
        "close": (None, [C.c_void_p]),
        "get": (C.c_uint32, [C.c_void_p, C.c_uint32, C.c_uint32]),
        "set": (C.c_uint32, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]),
        "incr": (C.c_uint32, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]),
        "decr": (C.c_uint32, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]),
        "rowlen": (C.c_uint32, [C.c_void_p, C.c_uint32]),
        "getrow": (C.c_uint32, [C.c_void_p, C.c_uint32, _u32p, C.c_size_t]),
        "apply": (None, [C.c_void_p, C.c_int, C.c_size_t, _u32p, _u32p, _u32p, _u32p]),
        "sum_get": (C.c_uint64, [C.c_void_p, C.c_size_t, _u32p, _u32p]),
        "num_rows": (C.c_uint64, [C.c_void_p]),
        "dir_size": (C.c_uint64, [C.c_void_p]),
        "mem": (C.c_uint64, [C.c_void_p]),
        "row_info": (C.c_int, [C.c_void_p, C.c_uint32, _u32p, _u32p]),
        "row_slots": (C.c_uint32, [C.c_void_p, C.c_uint32, _u32p, C.c_uint32]),
        "list_rows": (C.c_uint64, [C.c_void_p, _u32p, C.c_uint64]),
    }
    out = {}
    for short, (res, args) in sig.items():
        fn = getattr(lib, names[short])
        fn.restype, fn.argtypes = res, args
        out["_" + short] = staticmethod(fn)
    return out


def _make(cls_name, so, prefix_api, prefix_aux):
    lib = C.CDLL(so)
    api = ["open", "close", "get", "set", "incr", "decr", "rowlen", "getrow"]
    aux = ["apply", "sum_get", "num_rows", "dir_size", "mem", "row_info", "row_slots", "list_rows"]
    names = {n: prefix_api + n for n in api}
    names.update({n: prefix_aux + n for n in aux})
    return type(cls_name, (_Base,), dict(_bind(lib, names), _lib=lib))


_cache = {}


def Oracle(fname=None):
    if "o" not in _cache:
        if not os.path.exists(ORACLE_SO):
            build()
        _cache["o"] = _make("Oracle", ORACLE_SO, "ora_", "ora_")
    return _cache["o"](fname)


def cf_neighbors(oracle_handle, item, cap):
    """ora_cf_neighbors on an Oracle instance -> (ids, scores)"""
    lib = oracle_handle._lib
    fn = lib.ora_cf_neighbors
    fn.restype = C.c_uint32
    fn.argtypes = [C.c_void_p, C.c_uint32, _u32p, C.POINTER(C.c_double), C.c_uint32]
    ids = np.zeros(max(cap, 1), dtype=np.uint32)
    sc = np.zeros(max(cap, 1), dtype=np.float64)
    n = fn(oracle_handle._h, item, _ptr(ids), sc.ctypes.data_as(C.POINTER(C.c_double)), cap)
    return ids[:n], sc[:n]


def cf_import_preference_set(oracle_handle, ids):
    """ora_cf_import_preference_set on an Oracle instance (examples/cf_recommender.c:36-47)"""
    fn = oracle_handle._lib.ora_cf_import_preference_set
    fn.restype = None
    fn.argtypes = [C.c_void_p, _u32p, C.c_uint32]
    ids = np.ascontiguousarray(ids, dtype=np.uint32)
    fn(oracle_handle._h, _ptr(ids), ids.size)


def Reference(fname=None):
    if "r" not in _cache:
        if not have_reference():
            raise FileNotFoundError(REF_SO + " (run make -C oracle where /root/reference exists)")
        _cache["r"] = _make("Reference", REF_SO, "smatrix_", "ref_")
    return _cache["r"](fname)
