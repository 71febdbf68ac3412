#!/bin/bash
# Runs on the GPU box: the bench line, the rocprofv3 kernel stats of the same command and the PMC passes (each in its
# own run, counters only with --kernel-trace), for config 2 (bench.py) and config 3 (bench.py --config 3), condensed
# into gpurun_out/prof_*.  tools/build_profiles.py then writes the files committed under profiles/.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
# (bench.py prints the SHORT line; the full result -- what tools/build_profiles.py reads -- is bench_detail.json)
python3 $R/bench.py > $O/prof_bench_line.json 2> $O/prof_bench.err
cp $R/bench_detail.json $O/prof_bench.json
# config 2: kernel stats + spans + timeline
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --no-cpu --no-extras > /dev/null 2>&1
cp $R/bench_detail.json $O/prof_bench_under_rocprof.json
python3 $R/tools/prof_summary.py $O/kt $O/prof_kernel_stats.txt
python3 $R/tools/step_spans.py $O/kt > $O/prof_step_spans.txt
python3 $R/tools/timeline.py $O/kt last-growing > $O/prof_timeline_step23.txt
rm -rf $O/kt
# config 3: kernel stats of the getrow scan
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt3 -- python3 $R/bench.py --config 3 > /dev/null 2>&1
cp $R/bench_detail_config3.json $O/prof_bench_config3_under_rocprof.json
python3 $R/tools/prof_summary.py $O/kt3 $O/prof_kernel_stats_config3.txt
rm -rf $O/kt3
# PMC passes (one counter set per run)
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum TCC_EA0_ATOMIC_sum"; do
  tag=$(echo $c | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$tag -- python3 $R/bench.py --no-cpu --no-extras > /dev/null 2>&1
done
python3 $R/tools/pmc_summary.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_TCC_HIT_sum > $O/prof_pmc_raw.txt
rm -rf $O/pmc_*
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc3_$c -- python3 $R/bench.py --config 3 > /dev/null 2>&1
done
python3 $R/tools/pmc_summary.py $O/pmc3_FETCH_SIZE $O/pmc3_WRITE_SIZE > $O/prof_pmc_raw_config3.txt
rm -rf $O/pmc3_*
ls -la $O/prof_*
