#!/bin/bash
# Runs on the GPU box: the dense-id stream (tools/probe/dense_steps.py 24 = bench.py's dense_ids leg) under rocprofv3 -- kernel trace
# (time by kernel over the last 8 steps and over steps 2..5) and the PMC passes (each counter set in its own run, counters only with
# --kernel-trace) -- condensed into gpurun_out/prof_dense_*.  tools/build_dense_profile.py then writes profiles/r06_dense_kernels.txt
# and profiles/pmc_dense.json (keyed to the kernel source hash: bench.py's `roofline_dense` reads it).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O; rm -rf $O/kt_dense $O/pmcd_*
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/kt_dense -- python3 $R/tools/probe/dense_steps.py 24 > $O/prof_dense_run.txt 2>/dev/null
python3 $R/tools/probe/kernel_sums_window.py $O/kt_dense 8 > $O/prof_dense_kernels_last8.txt
python3 $R/tools/probe/kernel_sums_window.py $O/kt_dense 22 4 > $O/prof_dense_kernels_steps2to5.txt
python3 $R/tools/probe/step_kernels.py $O/kt_dense | tail -2 > $O/prof_dense_sequence.txt
rm -rf $O/kt_dense
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $c | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmcd_$tag -- python3 $R/tools/probe/dense_steps.py 24 > /dev/null 2>&1
  for k in k_apply_wpo_far k_far_scan k_grow_rest_lds "k_apply_agg_clu<2>" k_get_clu k_far_rows k_far_keys k_far_absent k_far_place k_pend_group k_prep; do
    python3 $R/tools/probe/pmc_kernel.py $O/pmcd_$tag "$k"
  done > $O/prof_dense_pmc_$tag.txt
  rm -rf $O/pmcd_$tag
done
ls -la $O/prof_dense_*
