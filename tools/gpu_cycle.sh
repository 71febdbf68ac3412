#!/bin/bash
# one GPU-box cycle: parity tests, two bench lines (metric only), kernel timeline of a late step (-> gpurun_out/)
#   gpurun --timeout 3000 -- 'bash tools/gpu_cycle.sh [notest]'
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
if [ "$1" != "notest" ]; then timeout 2400 python -m pytest tests -m gpu -q -x --durations=6 2>&1 | tail -9 | cut -c1-300; fi
P='import json,sys; d=json.loads(sys.stdin.read()); print("value %.0f Mops/s  ms/step %.3f  incr %.3f ms  get %.3f ms  rounds %d" % (d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline_get"]["avg_launch_ms"], d["table"]["rounds"]))'
for i in 1 2; do python bench.py --no-cpu --no-extras 2>/dev/null | python -c "$P"; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/kt -- python3 $R/bench.py --no-cpu --no-extras --no-profile 2>/dev/null | python3 -c "$P"
cd $R
python tools/step_spans.py gpurun_out/kt | tail -8 > gpurun_out/spans.txt
python tools/timeline.py gpurun_out/kt last-growing > gpurun_out/tl23.txt
rm -rf gpurun_out/kt
