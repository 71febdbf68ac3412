"""SparseMatrix -- the host-side mirror of the reference's binding interface.

Same method names, argument meaning and error behaviour as the reference's Java
class (src/java/com/paulasmuth/libsmatrix/SparseMatrix.java:70-118) and Ruby class
(src/smatrix_ruby.c:166-174), over the drop-in C ABI (include/smatrix.h), plus the
batched calls (include/smatrix_batch.h) that feed the HIP kernels.  Everything
runs on the GPU through lib/smatrix.so; there is no CPU path here.
"""
import ctypes as C

import numpy as np

from . import _lib

OP_GET, OP_SET, OP_INCR, OP_DECR = 0, 1, 2, 3


def _u32(a):
    return np.ascontiguousarray(a, dtype=np.uint32)


def _p(a):
    return a.ctypes.data_as(_lib.u32p)


class SparseMatrix:
    def __init__(self, filename=None):
        """filename None = in-memory (SparseMatrix.java:70-77); else open-or-create the file."""
        self._lib = _lib.load()
        self.filename = filename
        self._h = self._lib.smatrix_open(filename.encode() if filename else None)
        if not self._h:
            # smatrix_jni.c:61-62 turns a NULL handle into IllegalArgumentException
            raise ValueError("smatrix_open() failed (no HIP device, or cannot open %r)" % (filename,))

    # ---- the reference binding's methods ---------------------------------
    def get(self, x, y):
        return self._lib.smatrix_get(self._h, x, y)

    def set(self, x, y, v):
        return self._lib.smatrix_set(self._h, x, y, v)

    def incr(self, x, y, v):
        return self._lib.smatrix_incr(self._h, x, y, v)

    def decr(self, x, y, v):
        return self._lib.smatrix_decr(self._h, x, y, v)

    def getRowLength(self, x):
        return self._lib.smatrix_rowlen(self._h, x)

    def getRow(self, x, maxlen=0):
        """SortedMap<Integer,Integer> in Java (SparseMatrix.java:102-112): a dict sorted by key.
        Buffer sizing and the maxlen cut follow smatrix_jni.c:130-144."""
        pairs = self.getrow_raw(x, self.getRowLength(x) * 8)
        if maxlen > 0:
            pairs = pairs[:maxlen]
        return dict(sorted((int(k), int(v)) for k, v in pairs))

    def getFilename(self):
        return self.filename

    def close(self):
        if self._h:
            self._lib.smatrix_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- C-ABI level helpers ---------------------------------------------
    def getrow_raw(self, x, ret_len_bytes):
        """smatrix_getrow with a caller buffer of ret_len_bytes: (n,2) pairs in slot order."""
        buf = np.zeros(ret_len_bytes // 4 + 4, dtype=np.uint32)   # slack: S4 may overrun by one pair
        n = self._lib.smatrix_getrow(self._h, x, _p(buf), ret_len_bytes)
        return buf[: 2 * n].reshape(n, 2).copy()

    @property
    def mem(self):
        return int(self._h.contents.mem)

    # ---- batched API --------------------------------------------------------
    def apply_batch(self, op, x, y, v=None, results=True, out=None):
        """results=False: no result array is passed (include/smatrix_batch.h: the table ends in the same state, the
        kernels skip the results) and None is returned; out: the caller's own result array (uint32, same length)"""
        x, y = _u32(x), _u32(y)
        if x.shape != y.shape:
            raise ValueError("x and y must have the same shape")
        if out is not None:
            # (explicit checks, not asserts: python -O strips those, and a raw pointer goes to C -- ADVICE r5)
            if not results:
                raise ValueError("out given with results=False")
            if not isinstance(out, np.ndarray) or out.dtype != np.uint32 or out.shape != x.shape or not out.flags["C_CONTIGUOUS"]:
                raise ValueError("out must be a C-contiguous uint32 array of the batch's shape")
        else:
            out = np.empty_like(x) if results else None
        vv = _u32(v) if v is not None else None
        self._lib.smatrix_apply_batch(self._h, op, x.size, _p(x), _p(y),
                                      _p(vv) if vv is not None else None, _p(out) if results else None)
        return out

    def get_batch(self, x, y, out=None):
        return self.apply_batch(OP_GET, x, y, out=out)

    def set_batch(self, x, y, v):
        return self.apply_batch(OP_SET, x, y, v)

    def incr_batch(self, x, y, v, out=None):
        return self.apply_batch(OP_INCR, x, y, v, out=out)

    def decr_batch(self, x, y, v):
        return self.apply_batch(OP_DECR, x, y, v)

    def rowlen_batch(self, xs):
        xs = _u32(xs)
        out = np.empty_like(xs)
        self._lib.smatrix_rowlen_batch(self._h, xs.size, _p(xs), _p(out))
        return out

    def getrow_batch(self, xs, caps=None):
        """returns (offsets[n+1], pairs[total,2], counts[n]); row r's pairs are
        pairs[offsets[r]:offsets[r]+counts[r]].  caps default: rowlen+1 (quirk Q5)."""
        xs = _u32(xs)
        if caps is None:
            caps = self.rowlen_batch(xs).astype(np.uint64) + 1
        offsets = np.zeros(xs.size + 1, dtype=np.uint64)
        np.cumsum(np.asarray(caps, dtype=np.uint64), out=offsets[1:])
        pairs = np.zeros((int(offsets[-1]), 2), dtype=np.uint32)
        counts = np.zeros(xs.size, dtype=np.uint32)
        self._lib.smatrix_getrow_batch(self._h, xs.size, _p(xs), offsets.ctypes.data_as(_lib.u64p),
                                       pairs.ctypes.data_as(_lib.u32p), _p(counts))
        return offsets, pairs, counts

    def cf_neighbors_batch(self, items, caps=None):
        """CF-recommender read path (examples/cf_recommender.c:50-86), fused on the GPU:
        -> (offsets[n+1], ids[total], scores[total] float64, counts[n])"""
        items = _u32(items)
        if caps is None:
            caps = self.rowlen_batch(items).astype(np.uint64) + 1
        offsets = np.zeros(items.size + 1, dtype=np.uint64)
        np.cumsum(np.asarray(caps, dtype=np.uint64), out=offsets[1:])
        ids = np.zeros(int(offsets[-1]), dtype=np.uint32)
        scores = np.zeros(int(offsets[-1]), dtype=np.float64)
        counts = np.zeros(items.size, dtype=np.uint32)
        self._lib.smatrix_cf_neighbors_batch(self._h, items.size, _p(items), offsets.ctypes.data_as(_lib.u64p),
                                             _p(ids), scores.ctypes.data_as(C.POINTER(C.c_double)), _p(counts))
        return offsets, ids, scores, counts

    def cf_topk_batch(self, items, k):
        """the k best neighbours per item (k <= 64): -> (ids[n,k], scores[n,k] float64, counts[n])"""
        items = _u32(items)
        ids = np.zeros((items.size, k), dtype=np.uint32)
        scores = np.zeros((items.size, k), dtype=np.float64)
        counts = np.zeros(items.size, dtype=np.uint32)
        if self._lib.smatrix_cf_topk_batch(self._h, items.size, _p(items), k, ids.ctypes.data_as(_lib.u32p),
                                           scores.ctypes.data_as(C.POINTER(C.c_double)), _p(counts)) != 0:
            raise ValueError("k must be 1..64")
        return ids, scores, counts

    def cf_topk_batch_dev(self, n, items_ptr, k, ids_ptr, scores_ptr, cnt_ptr, stream=None):
        self._lib.smatrix_cf_topk_batch_dev(self._h, n, items_ptr, k, ids_ptr, scores_ptr, cnt_ptr, stream)

    def cf_import_sessions(self, sessions):
        """CF-recommender write path (examples/cf_recommender.c:36-47): every session is a sequence of item ids; all
        their L*L incr ops are generated and applied on the GPU"""
        lens = np.array([len(s) for s in sessions], dtype=np.uint64)
        offsets = np.zeros(len(sessions) + 1, dtype=np.uint64)
        np.cumsum(lens, out=offsets[1:])
        ids = _u32(np.concatenate([np.asarray(s, dtype=np.uint32) for s in sessions]) if len(sessions) else np.zeros(0, np.uint32))
        if self._lib.smatrix_cf_import_sessions(self._h, len(sessions), offsets.ctypes.data_as(_lib.u64p), _p(ids)) != 0:
            raise ValueError("smatrix_cf_import_sessions")

    def cf_import_sessions_dev(self, n_sessions, off_ptr, ids_ptr, op_off_ptr, total_ops, stream=None):
        self._lib.smatrix_cf_import_sessions_dev(self._h, n_sessions, off_ptr, ids_ptr, op_off_ptr, total_ops, stream)

    # device-pointer flavours (raw pointers; stream = hipStream_t as int or None)
    def apply_batch_dev(self, op, n, x_ptr, y_ptr, v_ptr, out_ptr, stream=None):
        self._lib.smatrix_apply_batch_dev(self._h, op, n, x_ptr, y_ptr, v_ptr, out_ptr, stream)

    def apply_packed_dev(self, op, n, rec_ptr, width, out_ptr, stream=None):
        """n records {x,y} (width 2, get) or {x,y,v} (width 3) in ONE device array"""
        if self._lib.smatrix_apply_packed_dev(self._h, op, n, rec_ptr, width, out_ptr, stream):
            raise ValueError("smatrix_apply_packed_dev: bad width / op")

    def rowlen_batch_dev(self, n, x_ptr, out_ptr, stream=None):
        self._lib.smatrix_rowlen_batch_dev(self._h, n, x_ptr, out_ptr, stream)

    def getrow_batch_dev(self, n, x_ptr, off_ptr, ret_ptr, cnt_ptr, stream=None):
        self._lib.smatrix_getrow_batch_dev(self._h, n, x_ptr, off_ptr, ret_ptr, cnt_ptr, stream)

    def cf_neighbors_batch_dev(self, n, items_ptr, off_ptr, ids_ptr, scores_ptr, cnt_ptr, stream=None):
        self._lib.smatrix_cf_neighbors_batch_dev(self._h, n, items_ptr, off_ptr, ids_ptr, scores_ptr, cnt_ptr, stream)

    # ---- introspection --------------------------------------------------------
    def stats(self):
        st = _lib.Stats()
        self._lib.smatrix_stats_sz(self._h, C.byref(st), C.sizeof(st))
        out = {n: getattr(st, n) for n, t in _lib.Stats._fields_ if t is C.c_uint64 or t is C.c_double}
        for i, op in enumerate(("get", "set", "incr", "decr")):
            out["kernel_ms_" + op] = st.kernel_ms[i]
            out["kernel_launches_" + op] = st.kernel_launches[i]
            out["kernel_ops_" + op] = st.kernel_ops[i]
        return out

    def reserve(self, nbytes):
        """capacity hint: map at least nbytes of device memory for row tables now (include/smatrix_batch.h smatrix_reserve)"""
        self._lib.smatrix_reserve(self._h, int(nbytes))

    def flush(self):
        """file mode: dirty rows reach the backing file now (include/smatrix_batch.h smatrix_flush)"""
        self._lib.smatrix_flush(self._h)

    def compact(self):
        """file mode: rewrite the backing file without leaked blocks (include/smatrix_batch.h smatrix_compact)"""
        if self._lib.smatrix_compact(self._h) != 0:
            raise RuntimeError("smatrix_compact did nothing: it is experimental and needs SMATRIX_EXPERIMENTAL=1 in the environment")

    def profile(self, on=True):
        self._lib.smatrix_profile(self._h, int(on))

    def row_info(self, x):
        size, used = C.c_uint32(0), C.c_uint32(0)
        ok = self._lib.smatrix_row_info(self._h, x, C.byref(size), C.byref(used))
        return (size.value, used.value) if ok else None

    def row_slots(self, x):
        info = self.row_info(x)
        if info is None:
            return None
        kv = np.zeros(2 * info[0], dtype=np.uint32)
        self._lib.smatrix_row_slots(self._h, x, _p(kv), info[0])
        return kv.reshape(-1, 2)


def device_available():
    return bool(_lib.load().smatrix_device_available())
