#!/bin/bash
# kernel time by name of the dense-id stream's steady steps (rocprofv3 --kernel-trace): bash tools/probe/dense_profile.sh [steps] [last] [N M]...
# (every further pair N M: the first M of the last N steps, e.g. 22 1 = step 2 of a 24-step run)
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out; rm -rf $R/gpurun_out/kt_dense
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/kt_dense -- python3 $R/tools/probe/dense_steps.py ${1:-24} 2>/dev/null | tail -2
cd $R
python tools/probe/kernel_sums_window.py gpurun_out/kt_dense ${2:-8}
shift 2
while [ -n "$2" ]; do python tools/probe/kernel_sums_window.py gpurun_out/kt_dense $1 $2 | head -9; shift 2; done
python tools/probe/step_kernels.py gpurun_out/kt_dense | tail -2 | cut -c1-900
rm -rf gpurun_out/kt_dense
