#!/bin/bash
# the hand-run soaks of the final build (what profiles/r06_soak_runs.txt records): tests/cold_soak.py and tests/soak.py under the path switches
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for s in 71 72 73 74; do python tests/cold_soak.py 25 $s 2>&1 | tail -2 | head -1; done
SMATRIX_CLUSTERED=1 SMATRIX_COLD_MIN=1024 python tests/cold_soak.py 25 75 2>&1 | tail -2 | head -1
SMATRIX_PEND=0 python tests/cold_soak.py 12 76 2>&1 | tail -2 | head -1
SMATRIX_FAR_PLACE=0 python tests/cold_soak.py 12 77 2>&1 | tail -2 | head -1
SMATRIX_REST_SLICE=64 python tests/cold_soak.py 12 78 2>&1 | tail -2 | head -1
python tests/soak.py 80 81 2>&1 | tail -1
python tests/soak.py 80 82 2>&1 | tail -1
SMATRIX_CLUSTERED=1 SMATRIX_HINT_LG=6 SMATRIX_COLD_MIN=2048 SMATRIX_COLD_SHARE=1024 python tests/soak.py 80 83 2>&1 | tail -1
SMATRIX_SPEC_TINY=1 SMATRIX_BULK_MIN=256 SMATRIX_BULK_SHARE=64 python tests/soak.py 80 84 2>&1 | tail -1
SMATRIX_CLUSTERED=1 SMATRIX_REST_SLICE=64 python tests/soak.py 80 85 2>&1 | tail -1
# (two-entry hint slots: tables of 4 and 8 slots -- every put pushes an entry back or out)
SMATRIX_CLUSTERED=1 SMATRIX_HINT_LG=2 python tests/soak.py 80 86 2>&1 | tail -1
SMATRIX_CLUSTERED=1 SMATRIX_HINT_LG=3 SMATRIX_COLD_MIN=1024 python tests/cold_soak.py 25 87 2>&1 | tail -2 | head -1
