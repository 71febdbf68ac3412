#!/usr/bin/env python3
"""Generates tests/golden/*.json from the REAL reference (oracle/_ref).

Run in the build container only (needs /root/reference to have been compiled by
`make -C oracle`).  The fixtures are data: inputs (or the generator parameters
that define them) and the reference's outputs.  TEST INFRASTRUCTURE ONLY.

    python oracle/gen_golden.py            # rewrites tests/golden/*.json
"""
import hashlib
import json
import os
import struct
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import oracle as O  # noqa: E402
from libsmatrix_amd import stream as S  # noqa: E402  (the product's stream generator, host side)

GOLD = os.path.join(ROOT, "tests", "golden")


def layout_digest(m, xs=None):
    """sha256 over (x, size, used, raw slots) of every row, rows sorted by x."""
    h = hashlib.sha256()
    rows = sorted(m.list_rows().tolist()) if xs is None else sorted(xs)
    for x in rows:
        size, used = m.row_info(x)
        h.update(struct.pack("<III", x, size, used))
        h.update(m.row_slots(x).astype("<u4").tobytes())
    return h.hexdigest()


def content_digest(m):
    """sha256 over (x, size, used, sorted non-empty pairs): layout-independent."""
    h = hashlib.sha256()
    for x in sorted(m.list_rows().tolist()):
        size, used = m.row_info(x)
        kv = m.row_slots(x)
        ne = kv[(kv[:, 0] != 0) | (kv[:, 1] != 0)]
        ne = ne[np.lexsort((ne[:, 1], ne[:, 0]))]
        h.update(struct.pack("<III", x, size, used))
        h.update(ne.astype("<u4").tobytes())
    return h.hexdigest()


def summary(m):
    return {
        "rows": m.num_rows(),
        "layout_sha256": layout_digest(m),
        "content_sha256": content_digest(m),
    }


# ---------------------------------------------------------------------------
def gen_quirks():
    """SURVEY.md Appendix A.1 sequence, expected values from the reference."""
    m = O.Reference()
    t = []  # transcript: [op, args..., result]

    def do(op, *a):
        r = getattr(m, op)(*a)
        if isinstance(r, np.ndarray):
            r = r.tolist()
        t.append([op, list(a), r])
        return r

    def row(x):
        info = m.row_info(x)
        t.append(["row", [x], None if info is None else
                  {"size": info[0], "used": info[1], "slots": m.row_slots(x).tolist()}])

    do("get", 7, 7); do("rowlen", 7)
    t.append(["num_rows", [], m.num_rows()])                          # S1
    do("decr", 1, 2, 30); do("incr", 1, 2, 5); do("set", 1, 2, 17); do("get", 1, 2)   # S2
    do("incr", 2, 0, 1); do("rowlen", 2); row(2)                      # Q1
    do("incr", 2, 16, 1); do("incr", 2, 32, 1); row(2); do("rowlen", 2); do("get", 2, 0)
    for y in range(1, 13):                                            # S5
        do("incr", 3, y, 1)
        info = m.row_info(3)
        t.append(["row_info", [3], list(info)])
    for y in range(100, 110):                                         # Q2
        do("incr", 2, y, 1)
    row(2); do("rowlen", 2)
    do("set", 4, 5, 0); do("rowlen", 4); do("get", 4, 5); row(4)      # S3
    do("getrow", 3, 256); do("getrow", 3, 24); do("getrow", 3, 20); do("getrow", 999, 64)  # S4
    do("set", 5, 0, 0); do("rowlen", 5); row(5)                       # Q3
    do("incr", 6, 0, 7); do("decr", 6, 0, 7); row(6)                  # (0,v) back to empty
    for y in range(1, 10):
        do("set", 6, y * 16, y)                                       # collisions at home 0
    row(6)
    do("set", 6, 160, 1); row(6)                                      # 10th key: 16 -> 32
    # full 32-bit patterns round-trip (JNI passes jint reinterpret-cast, smatrix_jni.c:95-111)
    do("set", 0xFFFFFFFF, 0xFFFFFFFE, 0xFFFFFFFD); do("get", 0xFFFFFFFF, 0xFFFFFFFE)
    do("incr", 0xFFFFFFFF, 0xFFFFFFFE, 5); do("set", 0, 1, 9); do("get", 0, 1); do("rowlen", 0)
    t.append(["mem", [], m.mem()])
    t.append(["summary", [], summary(m)])
    m.close()
    return {"source": "reference compiled from /root/reference/src/smatrix.c", "transcript": t}


# ---------------------------------------------------------------------------
def gen_java_suite():
    """src/java/test/TestSparseMatrix.java:22-131 replayed at the C ABI on ONE shared
    in-memory handle, in the suite's order (cases see earlier cases' data)."""
    m = O.Reference()
    out = {}
    m.set(42, 23, 17); out["case1_get"] = m.get(42, 23)
    m.set(4231, 2634, 0); m.incr(4231, 2634, 1); out["case2_get"] = m.get(4231, 2634)
    m.set(1231, 2634, 0); m.incr(1231, 2634, 1); m.incr(1231, 2634, 5)
    out["case3_get"] = m.get(1231, 2634)
    n, i = np.meshgrid(np.arange(1000, dtype=np.uint32), np.arange(1000, dtype=np.uint32), indexing="ij")
    xs, ys = i.ravel(), n.ravel()          # for n: for i: set(i, n, 34)
    m.apply(O.OP_SET, xs, ys, np.full(xs.size, 34, np.uint32))
    got = m.apply(O.OP_GET, xs, ys)
    out["case4_all_34"] = bool((got == 34).all())
    r = np.arange(1000, dtype=np.uint32)
    m.apply(O.OP_INCR, r, np.full(1000, 42, np.uint32), np.ones(1000, np.uint32))
    out["case5_rowlen_42"] = m.rowlen(42)
    m.apply(O.OP_INCR, r, np.full(1000, 85, np.uint32), np.ones(1000, np.uint32))
    out["case6_rowlen_85"] = m.rowlen(85)
    out["case6_getrow_85_pairs"] = int(m.getrow(85, m.rowlen(85) * 8).shape[0])
    m.apply(O.OP_INCR, r, np.full(1000, 83, np.uint32), np.ones(1000, np.uint32))
    out["case7_getrow_83_pairs"] = int(m.getrow(83, m.rowlen(83) * 8).shape[0])
    out["case7_maxlen"] = 230   # truncation happens in the JNI loop, smatrix_jni.c:141-144
    out["row_85_first_pairs"] = m.getrow(85, 16 * 8).tolist()
    out["summary"] = summary(m)
    out["mem"] = m.mem()
    m.close()
    return out


# ---------------------------------------------------------------------------
STREAMS = [
    # name, dist, n_ids, scramble, n_small (per-op returns kept), n_big (checksums only)
    ("uniform_dense", "uniform", 1 << 12, 0, 4000, 300000),
    ("uniform_scrambled", "uniform", 1 << 20, 1, 4000, 300000),
    ("zipf_dense", "zipf", 100000, 0, 4000, 300000),
    ("zipf_scrambled", "zipf", 1000000, 1, 4000, 1000000),
]


def gen_streams():
    out = {"seed": 12345, "zipf_s": 1.1, "cases": []}
    for name, dist, n_ids, scr, n_small, n_big in STREAMS:
        gen = S.Stream(dist, 12345, n_ids, 1.1, scr)
        case = {"name": name, "dist": dist, "n_ids": n_ids, "scramble": scr}
        # small: full transcript
        x, y = gen.fill(0, n_small)
        m = O.Reference()
        ret = m.apply(O.OP_INCR, x, y, np.ones(n_small, np.uint32))
        case["small"] = {
            "n": n_small,
            "x_head": x[:8].tolist(), "y_head": y[:8].tolist(),
            "incr_returns": ret.tolist(),
            "sum_get": m.sum_get(x, y),
            "summary": summary(m),
        }
        # mixed ops on top: decr by 1 of every 3rd op, set of every 7th
        d = np.arange(0, n_small, 3)
        retd = m.apply(O.OP_DECR, x[d], y[d], np.ones(d.size, np.uint32))
        s = np.arange(0, n_small, 7)
        rets = m.apply(O.OP_SET, x[s], y[s], (s % 5).astype(np.uint32))
        case["small"]["decr_returns_sha256"] = hashlib.sha256(retd.astype("<u4").tobytes()).hexdigest()
        case["small"]["set_returns_sha256"] = hashlib.sha256(rets.astype("<u4").tobytes()).hexdigest()
        case["small"]["after_mixed"] = summary(m)
        case["small"]["after_mixed_sum_get"] = m.sum_get(x, y)
        m.close()
        # big: checksums only
        x, y = gen.fill(0, n_big)
        m = O.Reference()
        ret = m.apply(O.OP_INCR, x, y, np.ones(n_big, np.uint32))
        rows = m.list_rows()
        lens = np.array([m.rowlen(int(r)) for r in rows[:2000].tolist()], dtype=np.uint64)
        case["big"] = {
            "n": n_big,
            "incr_returns_sha256": hashlib.sha256(ret.astype("<u4").tobytes()).hexdigest(),
            "sum_get": m.sum_get(x, y),
            "rowlen_sum_first2000_rows_in_dir_order": int(lens.sum()),
            "summary": summary(m),
            "mem": m.mem(),
        }
        m.close()
        out["cases"].append(case)
    return out


# ---------------------------------------------------------------------------
def decode_file(path):
    """Own decoder of the reference's format (src/smatrix.c:30-72): returns a description."""
    with open(path, "rb") as f:
        data_len = os.fstat(f.fileno()).st_size
        hdr = f.read(512)
        desc = {"file_bytes": data_len, "magic": hdr[:8].hex(), "cmap_head_fpos": struct.unpack("<Q", hdr[8:16])[0],
                "header_rest_zero": hdr[16:] == bytes(496), "blocks": []}
        at = desc["cmap_head_fpos"]
        while at:
            f.seek(at)
            n, nxt = struct.unpack("<QQ", f.read(16))
            blk = {"fpos": at, "n_entries": n, "next": nxt, "entries": []}
            ents = f.read(n * 12)
            for i in range(n):
                x, fpos = struct.unpack_from("<IQ", ents, i * 12)
                if fpos == 0:
                    break
                f.seek(fpos)
                magic, size = struct.unpack("<8sQ", f.read(16))
                slots = np.frombuffer(f.read(size * 8), dtype="<u4").reshape(-1, 2)
                nz = np.nonzero((slots[:, 0] != 0) | (slots[:, 1] != 0))[0]
                blk["entries"].append({"x": x, "fpos": fpos, "magic": magic.hex(), "size": size,
                                       "slots": [[int(p), int(slots[p, 0]), int(slots[p, 1])] for p in nz]})
            desc["blocks"].append(blk)
            at = nxt
    return desc


def gen_fileformat():
    """SURVEY.md A.2 sequence written by the reference, decoded; then reopened by it."""
    path = os.path.join(tempfile.mkdtemp(prefix="smxgold"), "a2.smx")
    m = O.Reference(path)
    ops = [["set", 42, 23, 17], ["incr", 7, 0, 3], ["set", 9, 1, 5], ["set", 9, 17, 0], ["set", 9, 33, 6]]
    ops += [["incr", 3, y, y] for y in range(1, 13)]
    for op, *a in ops:
        getattr(m, op)(*a)
    before = {"get_9_33": m.get(9, 33), "rowlen_9": m.rowlen(9)}
    m.close()
    desc = decode_file(path)
    m = O.Reference(path)
    q = [["get", 42, 23], ["get", 7, 0], ["rowlen", 7], ["get", 9, 1], ["get", 9, 17], ["get", 9, 33],
         ["rowlen", 9], ["rowlen", 3], ["get", 3, 12], ["rowlen", 42], ["get", 1, 1], ["rowlen", 1]]
    after = [[op, a, getattr(m, op)(*a)] for op, *a in q]
    after_rows = m.num_rows()
    m.close()
    os.remove(path)
    return {"ops": ops, "before_close": before, "file": desc, "after_reopen": after,
            "rows_after_reopen": after_rows}


def main():
    if not O.have_reference():
        sys.exit("oracle/_ref is not built: run `make -C oracle` where /root/reference exists")
    os.makedirs(GOLD, exist_ok=True)
    for name, fn in [("quirks", gen_quirks), ("java_suite", gen_java_suite), ("streams", gen_streams),
                     ("fileformat", gen_fileformat)]:
        with open(os.path.join(GOLD, name + ".json"), "w") as f:
            json.dump(fn(), f, separators=(",", ":"))
        print("wrote", name, os.path.getsize(os.path.join(GOLD, name + ".json")), "bytes")


if __name__ == "__main__":
    main()
