#!/bin/bash
# round 2, first GPU cycle: scope probe, new tests, full GPU suite, bench line
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 tools/probe/atomic_scope.cpp -o /tmp/atomic_scope && timeout 300 /tmp/atomic_scope > gpurun_out/atomic_scope.txt 2>&1
df -h /tmp . > gpurun_out/df.txt 2>&1; nproc >> gpurun_out/df.txt; free -g >> gpurun_out/df.txt
timeout 2400 python -m pytest tests/test_gpu_configs.py -m gpu -q -x --durations=12 > gpurun_out/t_configs.txt 2>&1; tail -25 gpurun_out/t_configs.txt
timeout 2400 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_configs.py --durations=8 > gpurun_out/t_all.txt 2>&1; tail -15 gpurun_out/t_all.txt
timeout 1200 python bench.py > gpurun_out/bench1.json 2> gpurun_out/bench1.err; tail -c 3000 gpurun_out/bench1.err; cut -c1-600 gpurun_out/bench1.json
