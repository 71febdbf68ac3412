"""bench.py prints ONE short JSON line (the driver scans a bounded window; round 3's 20 KB line went unparsed).
The line builder is run here on round 3's full result and on a synthetic worst case."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _full():
    with open(os.path.join(ROOT, "profiles", "r03_bench.json")) as f:
        return json.load(f)


def test_short_line_is_short_and_complete():
    import bench
    full = _full()
    assert len(json.dumps(full)) > 8192                      # the input IS the oversized line
    text = bench.short_line(full, "bench_detail.json")
    assert len(text) < bench.LINE_MAX <= 4096 and "\n" not in text
    line = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["config"]["workload"] and "model" not in line["config"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in line["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in line["cpu_baseline"], k
    assert abs(line["roofline"]["frac"] - line["roofline"]["achieved"] / line["roofline"]["peak"]) < 1e-4
    assert abs(line["value"] - full["value"]) / full["value"] < 1e-3
    assert line["random_access"]["mixed_frac_of_read8"] and line["steady_state_all_hits"]["ms_per_step"]
    assert line["legs"]["config3"]["getrow_ms"] and line["legs"]["config5"]["verified"] is True


def test_short_line_sheds_optional_parts_before_it_overflows():
    import bench
    full = _full()
    full["op_kinds"]["Gops_per_s"] = {"kind_%d" % i: 1.0 / 3 for i in range(400)}      # an unexpectedly large leg
    full["config"]["workload"] = "w" * 5000
    text = bench.short_line(full, "bench_detail.json")
    assert len(text) < 4096
    line = json.loads(text)
    assert "roofline" in line and "cpu_baseline" in line and "value" in line


def test_sharded_line_carries_router_and_placement():
    import bench
    full = _full()
    full["n_gpus"] = 8
    full["config"].update({"router": "c", "placement": {"rows_placed_by_load": 256, "hash_range_widths": [0.1] * 8,
                                                         "ops_applied_over_mean": [1.0] * 8}})
    line = json.loads(bench.short_line(full))
    assert line["config"]["router"] == "c" and line["config"]["placement"]["ops_applied_over_mean"] == [1.0] * 8


def test_first_contact_watchdog_exits_with_one_line():
    """N > 1: a rank whose first contact with the others (process group, RCCL communicator, first routed batch) does not
    complete says why in ONE line and leaves with a non-zero code -- it never hangs and never re-execs"""
    import subprocess
    code = ("import sys, time; sys.path.insert(0, %r); import bench; "
            "bench.Watchdog(0.3, 5, 'first contact did not complete'); time.sleep(30)" % ROOT)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=25)
    assert p.returncode == 4
    lines = [ln for ln in p.stderr.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("bench.py rank 5: first contact did not complete within"), p.stderr


def test_self_launch_ends_the_siblings_of_a_failed_rank(tmp_path, monkeypatch):
    """`bench.py --gpus N` as a launcher: when one rank exits non-zero the others are ended (by PID) and the launcher returns
    that code instead of waiting for ranks that wait for a dead peer"""
    import bench
    import time
    script = tmp_path / "fake_rank.py"
    script.write_text("import os, sys, time\n"
                      "if os.environ['RANK'] == '1':\n"
                      "    print('rank 1: giving up', file=sys.stderr); sys.exit(3)\n"
                      "time.sleep(60)\n")
    monkeypatch.setattr(bench, "__file__", str(script))
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    t0 = time.time()
    assert bench.self_launch(3) == 3
    assert time.time() - t0 < 20
