"""CPU: the oracle (oracle/smatrix_oracle.c) against the golden vectors produced by the
real reference (oracle/gen_golden.py) and against SURVEY.md Appendix A."""
import hashlib
import os

import numpy as np

from libsmatrix_amd.stream import Stream
from tests import replay

OPS = {"get": 0, "set": 1, "incr": 2, "decr": 3}


def sha(a):
    return hashlib.sha256(np.asarray(a).astype("<u4").tobytes()).hexdigest()


def test_quirks_transcript(oracle_mod, golden):
    m = oracle_mod.Oracle()
    bad = replay.replay_quirks(m, golden("quirks")["transcript"], rows_of=lambda mm: mm.num_rows())
    mem = [w for op, a, w in golden("quirks")["transcript"] if op == "mem"][0]
    assert m.mem() == mem            # self->mem accounting, src/smatrix.c:151-166
    m.close()
    assert not bad, "\n".join(bad)


def test_survey_appendix_a1(oracle_mod):
    """values quoted in SURVEY.md A.1 (independent of the generated fixture)"""
    m = oracle_mod.Oracle()
    assert m.get(7, 7) == 0 and m.rowlen(7) == 0 and m.num_rows() == 0
    assert m.decr(1, 2, 30) == 4294967266 and m.incr(1, 2, 5) == 4294967271
    assert m.set(1, 2, 17) == 17 and m.get(1, 2) == 17
    assert m.incr(2, 0, 1) == 1 and m.rowlen(2) == 0
    m.incr(2, 16, 1); m.incr(2, 32, 1)
    assert m.row_info(2) == (16, 2) and m.row_slots(2)[:3].tolist() == [[0, 1], [16, 1], [32, 1]]
    seen = []
    for y in range(1, 13):
        m.incr(3, y, 1)
        seen.append(m.row_info(3))
    assert seen == [(16, k) for k in range(1, 10)] + [(32, 10), (32, 11), (32, 12)]
    for y in range(100, 110):
        m.incr(2, y, 1)
    assert m.row_info(2) == (32, 13)
    m.set(4, 5, 0)
    assert m.rowlen(4) == 1 and m.get(4, 5) == 0
    assert m.getrow(3, 24).shape[0] == 3 and m.getrow(3, 20).shape[0] == 3
    assert m.mem() == 1049536
    m.close()


def test_java_suite(oracle_mod, golden):
    """src/java/test/TestSparseMatrix.java:22-131 at the C ABI, one shared handle"""
    g = golden("java_suite")
    m = oracle_mod.Oracle()
    m.set(42, 23, 17); assert m.get(42, 23) == 17 == g["case1_get"]
    m.set(4231, 2634, 0); m.incr(4231, 2634, 1); assert m.get(4231, 2634) == 1 == g["case2_get"]
    m.set(1231, 2634, 0); m.incr(1231, 2634, 1); m.incr(1231, 2634, 5)
    assert m.get(1231, 2634) == 6 == g["case3_get"]
    n, i = np.meshgrid(np.arange(1000, dtype=np.uint32), np.arange(1000, dtype=np.uint32), indexing="ij")
    xs, ys = i.ravel(), n.ravel()
    m.apply(oracle_mod.OP_SET, xs, ys, np.full(xs.size, 34, np.uint32))
    assert (m.apply(oracle_mod.OP_GET, xs, ys) == 34).all() and g["case4_all_34"]
    r = np.arange(1000, dtype=np.uint32)
    m.apply(oracle_mod.OP_INCR, r, np.full(1000, 42, np.uint32), np.ones(1000, np.uint32))
    assert m.rowlen(42) == 1000 == g["case5_rowlen_42"]
    m.apply(oracle_mod.OP_INCR, r, np.full(1000, 85, np.uint32), np.ones(1000, np.uint32))
    assert m.rowlen(85) == g["case6_rowlen_85"]
    assert m.getrow(85, m.rowlen(85) * 8).shape[0] == 1000 == g["case6_getrow_85_pairs"]
    m.apply(oracle_mod.OP_INCR, r, np.full(1000, 83, np.uint32), np.ones(1000, np.uint32))
    assert m.getrow(83, m.rowlen(83) * 8).shape[0] == g["case7_getrow_83_pairs"]
    assert m.getrow(85, 16 * 8).tolist() == g["row_85_first_pairs"]
    rows = m.list_rows().tolist()
    assert len(rows) == g["summary"]["rows"]
    assert replay.layout_digest(m, rows) == g["summary"]["layout_sha256"]
    assert m.mem() == g["mem"]
    m.close()


def test_streams(oracle_mod, golden):
    g = golden("streams")
    for case in g["cases"]:
        gen = Stream(case["dist"], g["seed"], case["n_ids"], g["zipf_s"], case["scramble"])
        sm = case["small"]
        x, y = gen.fill(0, sm["n"])
        assert x[:8].tolist() == sm["x_head"] and y[:8].tolist() == sm["y_head"], case["name"]
        m = oracle_mod.Oracle()
        ret = m.apply(oracle_mod.OP_INCR, x, y, np.ones(sm["n"], np.uint32))
        assert ret.tolist() == sm["incr_returns"], case["name"]
        assert m.sum_get(x, y) == sm["sum_get"]
        rows = m.list_rows().tolist()
        assert replay.layout_digest(m, rows) == sm["summary"]["layout_sha256"], case["name"]
        d = np.arange(0, sm["n"], 3)
        assert sha(m.apply(oracle_mod.OP_DECR, x[d], y[d], np.ones(d.size, np.uint32))) == sm["decr_returns_sha256"]
        s = np.arange(0, sm["n"], 7)
        assert sha(m.apply(oracle_mod.OP_SET, x[s], y[s], (s % 5).astype(np.uint32))) == sm["set_returns_sha256"]
        assert replay.layout_digest(m, m.list_rows().tolist()) == sm["after_mixed"]["layout_sha256"]
        assert m.sum_get(x, y) == sm["after_mixed_sum_get"]
        m.close()

        big = case["big"]
        x, y = gen.fill(0, big["n"])
        m = oracle_mod.Oracle()
        ret = m.apply(oracle_mod.OP_INCR, x, y, np.ones(big["n"], np.uint32))
        assert sha(ret) == big["incr_returns_sha256"], case["name"]
        assert m.sum_get(x, y) == big["sum_get"]
        rows = m.list_rows().tolist()
        assert len(rows) == big["summary"]["rows"]
        assert replay.layout_digest(m, rows) == big["summary"]["layout_sha256"], case["name"]
        assert m.mem() == big["mem"]
        m.close()
        gen.close()


def test_survey_a4_checksum(oracle_mod):
    """SURVEY.md A.4 ([probe] on the reference): 10^6 scrambled Zipf(1.1) ops over 10^6 x 10^6
    ids -> 576561 nnz in 137116 rows; 10^6 uniform ops over 1000 x 1000 ids -> sum get 1999806"""
    gen = Stream("zipf", 12345, 1000000, 1.1, 1)
    x, y = gen.fill(0, 1000000)
    m = oracle_mod.Oracle()
    m.apply(oracle_mod.OP_INCR, x, y, np.ones(x.size, np.uint32))
    assert m.num_rows() == 137116
    assert int(m.apply(oracle_mod.OP_GET, x, y).astype(np.uint64).sum()) == m.sum_get(x, y)
    assert np.unique(x.astype(np.uint64) << 32 | y.astype(np.uint64)).size == 576561
    assert int(sum(m.rowlen(int(r)) for r in m.list_rows())) == 576561
    m.close()
    gen.close()
    gen = Stream("uniform", 12345, 1000, 1.1, 0)
    x, y = gen.fill(0, 1000000)
    m = oracle_mod.Oracle()
    m.apply(oracle_mod.OP_INCR, x, y, np.ones(x.size, np.uint32))
    assert m.num_rows() == 1000 and m.sum_get(x, y) == 1999806
    m.close()


def test_survey_a4_checksum_1e7(oracle_mod):
    """SURVEY.md A.4 ([probe] on the reference, 10^7 scrambled Zipf ops): sum over the stream of get(x_i,y_i) =
    52 480 898 544, 561 596 rows, 4 463 637 nnz, hottest row 159 472 -- the figures the GPU path is held to in
    tests/test_gpu_configs.py::test_config2_reference_checksums_1e7.  Where oracle/_ref exists, the compiled
    reference itself is run beside the oracle."""
    gen = Stream("zipf", 12345, 1000000, 1.1, 1)
    x, y = gen.fill(0, 10000000)
    for make in [oracle_mod.Oracle] + ([oracle_mod.Reference] if oracle_mod.have_reference() else []):
        m = make()
        m.apply(oracle_mod.OP_INCR, x, y, np.ones(x.size, np.uint32))
        assert m.sum_get(x, y) == 52480898544
        assert m.num_rows() == 561596
        lens = np.array([m.rowlen(int(r)) for r in m.list_rows()], dtype=np.uint64)
        assert int(lens.sum()) == 4463637 and int(lens.max()) == 159472
        m.close()
    gen.close()


def test_fileformat(oracle_mod, golden, tmp_path):
    """the oracle's writer/loader against the reference-written file's decoded description"""
    from oracle.gen_golden import decode_file
    g = golden("fileformat")
    path = str(tmp_path / "o.smx")
    m = oracle_mod.Oracle(path)
    for op, *a in g["ops"]:
        getattr(m, op)(*a)
    assert m.get(9, 33) == g["before_close"]["get_9_33"] and m.rowlen(9) == g["before_close"]["rowlen_9"]
    m.close()
    d, ref = decode_file(path), g["file"]
    assert d["magic"] == ref["magic"] == "17" * 8 and d["cmap_head_fpos"] == ref["cmap_head_fpos"] == 512
    assert d["header_rest_zero"] and len(d["blocks"]) == len(ref["blocks"]) == 1
    assert d["blocks"][0]["n_entries"] == ref["blocks"][0]["n_entries"] == 4194304
    assert d["blocks"][0]["fpos"] == ref["blocks"][0]["fpos"] == 512 and d["blocks"][0]["next"] == 0
    assert d["file_bytes"] == ref["file_bytes"] == 50332880
    mine = {e["x"]: (e["magic"], e["size"], e["slots"]) for e in d["blocks"][0]["entries"]}
    theirs = {e["x"]: (e["magic"], e["size"], e["slots"]) for e in ref["blocks"][0]["entries"]}
    assert mine == theirs                       # every row block byte-identical (position aside)
    assert [e["x"] for e in d["blocks"][0]["entries"]] == [e["x"] for e in ref["blocks"][0]["entries"]]
    m = oracle_mod.Oracle(path)
    for op, a, want in g["after_reopen"]:
        assert getattr(m, op)(*a) == want, (op, a)
    assert m.num_rows() == g["rows_after_reopen"]
    m.close()


def test_cf_example_restatement_known_answers():
    """examples/cf_recommender.c:36-47 + :50-86 as restated in oracle/smatrix_oracle.c, on a case small enough to do by
    hand: sessions (5,7,9) and (9,9).  Column 0 holds the totals; the pair test is on POSITIONS, so the doubled 9 counts
    itself; neighbours come in slot order with cc / (sqrt(total_a) * sqrt(total_b)), the (0,total) entry scoring 0."""
    from oracle import oracle as O
    o = O.Oracle()
    O.cf_import_preference_set(o, [5, 7, 9])
    O.cf_import_preference_set(o, [9, 9])
    assert [o.get(a, 0) for a in (5, 7, 9)] == [1, 1, 3]
    assert o.get(5, 7) == o.get(7, 5) == o.get(5, 9) == o.get(9, 5) == o.get(7, 9) == o.get(9, 7) == 1
    assert o.get(9, 9) == 2 and o.get(5, 5) == 0
    ids, sc = O.cf_neighbors(o, 9, 100)
    got = dict(zip(ids.tolist(), sc.tolist()))
    assert set(got) == {0, 5, 7, 9}
    assert got[0] == 0.0                                   # num 3 > den sqrt(3)*sqrt(1): the example's guard
    assert got[5] == 1.0 / (np.sqrt(3.0) * np.sqrt(1.0)) and got[7] == got[5]
    assert got[9] == 2.0 / (np.sqrt(3.0) * np.sqrt(3.0))
    o.close()
