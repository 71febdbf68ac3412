#!/bin/bash
# ops per wave of the wave-per-op kernels (SMX_WPO_OPS): rebuilds the library per value and runs the dense-id stream
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R/libsmatrix_amd/csrc
for k in ${@:-1 2 4 8}; do
  touch smx_runtime.hip
  make HIPCC="/opt/rocm/bin/hipcc -DSMX_WPO_OPS=$k" > /dev/null 2>&1
  echo -n "SMX_WPO_OPS=$k: "; (cd $R; python tools/probe/dense_steps.py 24 2>/dev/null | tail -1 | cut -c1-110)
done
touch smx_runtime.hip; make > /dev/null 2>&1
