#!/bin/bash
# agg kernel with parts left out (measurement builds): bash aggdbg.sh <lib> 
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/kt_dbg; cd /tmp && export TMPDIR=/tmp
export SMATRIX_LIB=$1 SMATRIX_DBG_AFTER=23
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/kt_dbg -- python3 $R/tools/probe/dense_steps.py 24 2>/dev/null | tail -2
cd $R
python tools/probe/kernel_sums_window.py gpurun_out/kt_dbg 4 2 | head -6
python tools/probe/kernel_sums_window.py gpurun_out/kt_dbg 2 | head -6
rm -rf gpurun_out/kt_dbg
