#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_configs.py -m gpu -q -x -k "native_router or incremental" 2>&1 | tail -5
P='import json,sys; d=json.loads(sys.stdin.read()); print("value %.0f ms/step %.3f" % (d["value"], d["ms_per_step"]))'
echo direct; timeout 600 python bench.py --no-cpu --no-extras 2>/dev/null | python -c "$P"
echo python-router; timeout 600 python bench.py --no-cpu --no-extras --force-sharded 2>gpurun_out/e1.txt | python -c "$P"
echo c-router; timeout 600 python bench.py --no-cpu --no-extras --force-sharded --c-router 2>gpurun_out/e2.txt | python -c "$P"; tail -3 gpurun_out/e2.txt
