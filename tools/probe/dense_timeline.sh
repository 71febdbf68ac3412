#!/bin/bash
# Runs on the GPU box: kernel timeline of the last step of the dense-id stream (start offsets, durations) -> gpurun_out/$1/timeline.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-tl}
mkdir -p $O; rm -rf $O/kt_dense
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/kt_dense -- python3 $R/tools/probe/dense_steps.py 24 > $O/run.txt 2>/dev/null
python3 $R/tools/timeline.py $O/kt_dense -1 > $O/timeline.txt
python3 $R/tools/probe/kernel_sums_window.py $O/kt_dense 8 > $O/kernels_last8.txt
rm -rf $O/kt_dense
tail -1 $O/run.txt | cut -c1-100; cat $O/timeline.txt
