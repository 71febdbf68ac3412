"""What the profile builders share: the hash that ties a committed PMC summary to the kernel sources it was taken with."""
import hashlib, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_source_sha16():
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "libsmatrix_amd", "csrc")
    for f in ["smx_kernels.hpp"] + sorted(os.path.join("kernels", k) for k in os.listdir(os.path.join(csrc, "kernels")) if k.endswith(".hpp")) + ["smx_runtime.hip"]:
        h.update(open(os.path.join(csrc, f), "rb").read())
    return h.hexdigest()[:16]
