#!/bin/bash
# one GPU-box cycle: parity tests, two bench lines, kernel timeline of a late step (-> gpurun_out/)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
if [ "$1" != "notest" ]; then timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -3 | cut -c1-400; fi
for i in 1 2; do python bench.py --no-cpu 2>/dev/null | grep "^{" | cut -c40-150; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/kt -- python3 $R/bench.py --no-cpu --no-profile 2>&1 | grep "^{" | cut -c40-150
cd $R
python tools/step_spans.py gpurun_out/kt | tail -8 > gpurun_out/spans.txt
python tools/timeline.py gpurun_out/kt 23 > gpurun_out/tl23.txt
rm -rf gpurun_out/kt
