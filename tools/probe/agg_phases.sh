#!/bin/bash
# Where does k_apply_agg<INCR>'s time go?  Measurement builds leave parts of phase 2 out after 14 normal batches
# (SMX_AGG_DBG: 1 hits without their atomic | 2 no global access in phase 2 | 3 directory loads only | 4 no slow path | 5 inserts without their `used` ticket).
# Build here (hipcc cross-compiles): tools/probe/agg_phases.sh build ; run on the GPU box: tools/probe/agg_phases.sh run
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R/libsmatrix_amd/csrc
V=../lib/variants
if [ "$1" = build ]; then
  mkdir -p $V
  for k in 1 2 3 4 5; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DSMX_AGG_DBG=$k -c smx_runtime.hip -o $V/rt_dbg$k.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $V/smatrix_dbg$k.so $V/rt_dbg$k.o ../lib/smx_stream.o -lm -pthread && rm $V/rt_dbg$k.o
  done
  ls -la $V
else
  cd $R
  echo "variant  incr_kernel_ms  ms_per_step"
  for k in 0 1 2 3 4 5; do
    L=$R/libsmatrix_amd/lib/smatrix.so; [ $k != 0 ] && L=$R/libsmatrix_amd/lib/variants/smatrix_dbg$k.so
    SMATRIX_LIB=$L SMATRIX_DBG_AFTER=15 python bench.py --no-cpu --no-extras --warmup 14 --steps 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('dbg$k   %.3f   %.3f' % (d['roofline']['avg_launch_ms'], d['ms_per_step']))"
  done
fi
