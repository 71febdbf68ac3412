#!/bin/bash
# GPU cycle: full GPU suite + bench line (no CPU leg) -> gpurun_out/
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 3000 python -m pytest tests -m gpu -q -x --durations=8 > gpurun_out/t_all.txt 2>&1; tail -12 gpurun_out/t_all.txt
timeout 1200 python bench.py --no-cpu $BENCH_ARGS > gpurun_out/bench2.json 2> gpurun_out/bench2.err; tail -c 1500 gpurun_out/bench2.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench2.json'))
print("value %.0f Mops/s  ms/step %.3f" % (d["value"], d["ms_per_step"]))
for k in ("sustained","dense_ids","steady_state_all_hits"):
    print(k, json.dumps(d.get(k))[:400])
print("incr kernel ms", d["roofline"]["avg_launch_ms"], "get", d["roofline_get"]["avg_launch_ms"])
c3=d.get("config3_getrow",{}); print("config3", c3.get("getrow_ms"), c3.get("build_Gops_per_s"), str(c3.get("error"))[:300])
c5=d.get("config5_file_1gpu",{}); print("config5", c5.get("close_s"), c5.get("open_s"), c5.get("verified"), str(c5.get("error"))[:300])
PY
