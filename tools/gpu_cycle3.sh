#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 3000 python -m pytest tests -m gpu -q -x --durations=6 > gpurun_out/t_all.txt 2>&1; tail -9 gpurun_out/t_all.txt
export SMATRIX_HIP_LIB=$R/libsmatrix_amd/lib/smatrix.so
(echo "# oracle/_ref/smatrix_benchmark_hip full  (the reference's UNCHANGED src/smatrix_benchmark.c against include/smatrix.h + lib/smatrix.o, scalar ABI, MI355X)"; timeout 600 oracle/_ref/smatrix_benchmark_hip full; echo "# reference itself, same box (tests/stock_benchmark_reference.py):"; timeout 600 python tests/stock_benchmark_reference.py) > gpurun_out/stock_table.txt 2>&1; cat gpurun_out/stock_table.txt
for i in 1 2; do timeout 600 python bench.py --no-cpu --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('value %.0f ms/step %.3f incr %.3f get %.3f' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline_get']['avg_launch_ms']))"; done
