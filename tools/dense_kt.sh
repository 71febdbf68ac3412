#!/bin/bash
# Runs on the GPU box: the kernel-trace half of tools/refresh_dense_profile.sh only (kernel time by name over the last 8 steps of
# the dense-id stream and the kernel sequence of its last two incr batches) into gpurun_out/$1/.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-kt}
mkdir -p $O; rm -rf $O/kt_dense
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/kt_dense -- python3 $R/tools/probe/dense_steps.py 24 > $O/run.txt 2>/dev/null
python3 $R/tools/probe/kernel_sums_window.py $O/kt_dense 8 > $O/kernels_last8.txt
python3 $R/tools/probe/kernel_sums_window.py $O/kt_dense 22 4 > $O/kernels_steps2to5.txt
python3 $R/tools/probe/step_kernels.py $O/kt_dense | tail -2 > $O/sequence.txt
rm -rf $O/kt_dense
tail -1 $O/run.txt | cut -c1-200
head -12 $O/kernels_last8.txt
