#!/bin/bash
# Runs on the GPU box: the bench line, the rocprofv3 kernel stats of the same command and the three PMC
# passes (each in its own run, counters only with --kernel-trace), condensed into gpurun_out/prof_*.
# Copy the results into profiles/ afterwards (tools/refresh_profiles.sh is the provenance of those files).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/prof_bench.json 2> $O/prof_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --no-cpu > $O/prof_bench_under_rocprof.json 2>/dev/null
python3 $R/tools/prof_summary.py $O/kt $O/prof_kernel_stats.txt
python3 $R/tools/step_spans.py $O/kt > $O/prof_step_spans.txt
python3 $R/tools/timeline.py $O/kt 23 > $O/prof_timeline_step23.txt
rm -rf $O/kt
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum TCC_EA0_ATOMIC_sum"; do
  tag=$(echo $c | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$tag -- python3 $R/bench.py --no-cpu > /dev/null 2>&1
done
python3 $R/tools/pmc_summary.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_TCC_HIT_sum > $O/prof_pmc_raw.txt
rm -rf $O/pmc_*
