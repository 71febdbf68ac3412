"""CPU, world_size 2, gloo: the multi-GPU routing path (partition -> all_to_all -> local apply ->
all_to_all back -> un-permute) of libsmatrix_amd/sharded.py against a single-matrix oracle."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("packed,split", [("0", "0"), ("1", "0"), ("1", "1")])
def test_sharded_two_ranks_gloo(packed, split):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", SMX_TEST_PACKED=packed, SMX_TEST_SPLIT=split)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(29533 + int(packed) + 2 * int(split)),
           os.path.join(ROOT, "tests", "sharded_worker.py")]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    assert "SHARDED_OK world=2" in p.stdout
