"""Shared replay logic: drives ANY implementation exposing the oracle-style methods
(get/set/incr/decr/rowlen/getrow/row_info/row_slots/apply/...) through the golden
transcripts and returns mismatches.  Used for the oracle (CPU tests) and for the HIP
library through its C ABI (GPU tests) -- the tests read like the reference's own."""
import hashlib
import struct

import numpy as np


def layout_digest(m, rows):
    h = hashlib.sha256()
    for x in sorted(rows):
        size, used = m.row_info(x)
        h.update(struct.pack("<III", x, size, used))
        h.update(np.asarray(m.row_slots(x)).astype("<u4").tobytes())
    return h.hexdigest()


def content_digest(m, rows):
    h = hashlib.sha256()
    for x in sorted(rows):
        size, used = m.row_info(x)
        kv = np.asarray(m.row_slots(x))
        ne = kv[(kv[:, 0] != 0) | (kv[:, 1] != 0)]
        ne = ne[np.lexsort((ne[:, 1], ne[:, 0]))]
        h.update(struct.pack("<III", x, size, used))
        h.update(ne.astype("<u4").tobytes())
    return h.hexdigest()


def replay_quirks(m, transcript, rows_of=None):
    """returns list of human-readable mismatches"""
    bad = []
    touched = set()
    for op, args, want in transcript:
        if op in ("get", "set", "incr", "decr", "rowlen"):
            got = getattr(m, op)(*args)
            if op != "get" and op != "rowlen":
                touched.add(args[0])
        elif op == "getrow":
            got = np.asarray(m.getrow(*args)).tolist()
        elif op == "row":
            info = m.row_info(*args)
            got = None if info is None else {"size": info[0], "used": info[1],
                                             "slots": np.asarray(m.row_slots(*args)).tolist()}
        elif op == "row_info":
            got = list(m.row_info(*args))
        elif op == "num_rows":
            got = len(touched) if rows_of is None else rows_of(m)
        elif op == "mem":
            continue   # allocator accounting is implementation-specific (checked for the oracle only)
        elif op == "summary":
            got = {"rows": len(touched), "layout_sha256": layout_digest(m, touched),
                   "content_sha256": content_digest(m, touched)}
        else:
            raise AssertionError(op)
        if got != want:
            bad.append("%s%r: got %r want %r" % (op, tuple(args), got, want))
    return bad
