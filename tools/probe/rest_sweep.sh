#!/bin/bash
# k_grow_rest_lds: slice length / grid sweep on the dense-id stream (mean step, ms), then the phase clocks of the default
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for sl in 256 512 1024 2048; do for g in 256 512 768 1536; do
  echo -n "slice $sl grid $g: "; SMATRIX_REST_SLICE=$sl SMATRIX_REST_GRID=$g python tools/probe/dense_steps.py 24 2>/dev/null | tail -1 | cut -c1-60
done; done
SMATRIX_REST_DBG=2 SMATRIX_REST_DBG_FROM=16 python tools/probe/dense_steps.py 24 2>&1 | grep "k_grow_rest"
