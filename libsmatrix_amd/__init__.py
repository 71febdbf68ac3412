"""libsmatrix_amd -- MI355X-native (gfx950) implementation of libsmatrix's
(x,y) -> uint32 get/set/incr/decr/getrow/rowlen path.

  SparseMatrix   mirror of the reference's Java/Ruby binding class over the drop-in C ABI
  Stream         deterministic uniform / Zipf op streams (host and on-device)
  ShardedMatrix  row-hash sharding over the GPUs of a node (torch.distributed / RCCL)

The compute path is lib/smatrix.so (HIP); importing works without a GPU, opening a
matrix does not.
"""
from .matrix import OP_DECR, OP_GET, OP_INCR, OP_SET, SparseMatrix, device_available  # noqa: F401
from .stream import Stream  # noqa: F401
