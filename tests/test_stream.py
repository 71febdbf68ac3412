"""The synthetic stream generator (include/smx_stream.h): host implementation against the specification's
known values (CPU), and the on-device generator against the host one, bit for bit (GPU)."""
import numpy as np
import pytest

from libsmatrix_amd import _lib
from libsmatrix_amd.stream import Stream

CF_200K_ROWS_NNZ = 22999896


def splitmix_ref(seed, n):
    """SplitMix64 as specified in SURVEY.md Appendix B (sequential form)"""
    M = (1 << 64) - 1
    out, state = [], seed
    for _ in range(n):
        state = (state + 0x9E3779B97F4A7C15) & M
        z = state
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
        out.append(z ^ (z >> 31))
    return out


def test_splitmix_random_access_equals_sequential():
    lib = _lib.load()
    want = splitmix_ref(12345, 50)
    assert [lib.smx_splitmix64_at(12345, j) for j in range(50)] == want


def test_uniform_and_scramble_definition():
    lib = _lib.load()
    g = Stream("uniform", 12345, 1 << 20, 1.1, 0)
    x, y = g.fill(0, 1000)
    r = splitmix_ref(12345, 2000)
    assert x.tolist() == [1 + (r[2 * i] % (1 << 20)) for i in range(1000)]
    assert y.tolist() == [1 + (r[2 * i + 1] % (1 << 20)) for i in range(1000)]
    gs = Stream("uniform", 12345, 1 << 20, 1.1, 1)
    xs, _ = gs.fill(0, 1000)
    assert xs.tolist() == [lib.smx_fmix32(int(v)) for v in x]
    # windows of the stream are consistent with each other (random access by op index)
    x2, y2 = g.fill(400, 100)
    assert (x2 == x[400:500]).all() and (y2 == y[400:500]).all()


def test_zipf_inverse_cdf_definition():
    g = Stream("zipf", 12345, 1000, 1.1, 0)
    n = 5000
    x, y = g.fill(0, n)
    w = np.arange(1, 1001, dtype=np.float64) ** -1.1
    cdf = np.cumsum(w) / w.sum()
    r = splitmix_ref(12345, 2 * n)
    u = np.array([(v >> 11) * 2.0 ** -53 for v in r])
    rank = 1 + np.searchsorted(cdf, u, side="left")
    # libm pow vs numpy power may differ in the last ulp at a bin edge: allow a handful of off-by-one ranks
    assert (np.abs(rank[0::2].astype(np.int64) - x.astype(np.int64)) <= 1).all()
    assert (rank[0::2] != x).sum() + (rank[1::2] != y).sum() <= 3
    assert abs((x == 1).mean() - cdf[0]) < 0.02            # P(rank 1) = 1/H ~ 0.13 for N = 1000


def test_cf_shape_definition():
    """SMX_DIST_CF (BASELINE config 3): op i -> row fmix32(1 + i // per_row), column
    fmix32(1 + draw_i % n_cols), ONE draw per op"""
    lib = _lib.load()
    g = Stream("cf", 12345, 13000000, 115.0, 1)
    x, y = g.fill(0, 1000)
    r = splitmix_ref(12345, 1000)
    assert x.tolist() == [lib.smx_fmix32(1 + i // 115) for i in range(1000)]
    assert y.tolist() == [lib.smx_fmix32(1 + r[i] % 13000000) for i in range(1000)]
    x2, y2 = g.fill(115 * 7 + 3, 300)                     # random access by op index
    x3, y3 = g.fill(0, 115 * 7 + 303)
    assert (x2 == x3[-300:]).all() and (y2 == y3[-300:]).all()
    # distinct (row, column) cells of the first 200 000 rows; the full 13 M-row stream holds 1 494 993 467 of its
    # 1 495 000 000 ops as distinct cells (same count, run over all rows: tests/test_gpu_configs.py::CF13M_NNZ)
    x, y = g.fill(0, 200000 * 115)
    assert np.unique(x.astype(np.uint64) << 32 | y).size == CF_200K_ROWS_NNZ
    gd = Stream("cf", 5, 1000, 3.0, 0)
    xd, yd = gd.fill(0, 9)
    assert xd.tolist() == [1, 1, 1, 2, 2, 2, 3, 3, 3] and yd.min() >= 1 and yd.max() <= 1000


@pytest.mark.gpu
@pytest.mark.parametrize("dist,n_ids,scr", [("zipf", 1000000, 1), ("zipf", 5000, 0), ("uniform", 1 << 20, 1),
                                            ("cf", 13000000, 1)])
def test_device_generator_equals_host(dist, n_ids, scr):
    import torch
    g = Stream(dist, 777, n_ids, 115.0 if dist == "cf" else 1.1, scr)
    first, n = 123456789, 300000
    hx, hy = g.fill(first, n)
    dx = torch.empty(n, dtype=torch.int32, device="cuda"); dy = torch.empty_like(dx)
    g.fill_device(first, n, dx.data_ptr(), dy.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert (dx.cpu().numpy().view(np.uint32) == hx).all() and (dy.cpu().numpy().view(np.uint32) == hy).all()
