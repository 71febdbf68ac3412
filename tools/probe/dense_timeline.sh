#!/bin/bash
# Runs on the GPU box: kernel timeline of one step of the dense-id stream (start offsets, durations) -> gpurun_out/$1/timeline*.txt
#   $2: which step of the trace (tools/timeline.py: index among the full-grid launches of the folding kernel; default -1 = the last)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-tl}
mkdir -p $O; rm -rf $O/kt_dense
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/kt_dense -- python3 $R/tools/probe/dense_steps.py 24 > $O/run.txt 2>/dev/null
for w in ${2:--1} $3 $4; do python3 $R/tools/timeline.py $O/kt_dense $w > $O/timeline_$w.txt; done
python3 $R/tools/probe/kernel_sums_window.py $O/kt_dense 8 > $O/kernels_last8.txt
python3 $R/tools/probe/kernel_sums_window.py $O/kt_dense 22 4 > $O/kernels_steps2to5.txt
rm -rf $O/kt_dense
tail -1 $O/run.txt | cut -c1-100; cat $O/timeline_${2:--1}.txt
