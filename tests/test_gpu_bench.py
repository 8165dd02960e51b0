"""bench.py end to end on the GPU box: the driver reads ONE JSON line from the tail of stdout."""
import json
import subprocess
import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu


def _run(args, timeout=600):
    proc = subprocess.run([sys.executable, str(REPO / "bench.py")] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                          cwd=str(REPO), timeout=timeout)
    assert proc.returncode == 0, proc.stderr[-3000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, "stdout holds the headline line and nothing else: %r" % [ln[:80] for ln in lines]
    assert len(lines[0]) < 4096
    return json.loads(lines[0]), proc.stderr


def test_default_command_prints_one_short_headline_line():
    """`python bench.py` (the configs[] entries cut short by the budget): the line is the headline alone, under 4 KB, with
    `roofline.frac` and `cpu_baseline.value` in it; every configuration that ran has its own `[bench-config]` stderr line."""
    j, err = _run(["--steps", "8", "--warmup", "2", "--cpu-seconds", "2", "--budget-seconds", "1"])
    assert j["metric"] == "co-occurrence nonzeros/sec" and j["unit"] == "nonzeros/s" and j["n_gpus"] == 1 and j["steps"] == 8
    assert j["config"]["workload"] == "zipf_v400k_d300" and j["config"]["index"] == "dealt" and j["dtype"] == "f32"
    rf, cb = j["roofline"], j["cpu_baseline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and 0.0 < rf["frac"] < 1.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    assert abs(rf["achieved"] - rf["algorithmic_bytes_per_step"] / (j["ms_per_step"] * 1e-3) / 1e9) < 1e-3 * rf["achieved"]
    assert rf["traffic"] > rf["algorithmic_bytes_per_step"] and rf["traffic_source"].startswith("profiles/")
    assert cb["value"] > 0 and cb["cores"] >= 1 and cb["kind"] == "port" and "Adagrad" in cb["sample"]
    assert abs(j["value"] - 1048576 / (j["ms_per_step"] * 1e-3)) < 1e-6 * j["value"]
    assert len(j["configs_skipped"]) == 12 and j["configs_run"] == []
    marks = [ln for ln in err.splitlines() if ln.startswith("[bench-config] ")]
    assert len(marks) == 1 and json.loads(marks[0][len("[bench-config] "):])["name"] == "headline"
    assert all(len(ln) < 1024 for ln in marks)


def test_two_ranks_rehearsed_on_one_gpu_print_one_short_headline_line():
    """`python bench.py --gpus 2` starts its ranks itself; rehearsed here with both ranks on cuda:0 over gloo (control flow only).
    The headline runs in the trainer's default mode (epochs dealt, both tables sharded) and carries the process group and the
    collectives' share."""
    j, _ = _run(["--gpus", "2", "--rehearse-on-one-gpu", "--steps", "4", "--warmup", "1"])
    assert j["n_gpus"] == 2 and j["config"]["index"] == "dealt" and j["config"]["global_batch"] == 2 * 1048576
    assert "both tables sharded x2" in j["config"]["parallelism"] and "rehearsal" in j["config"]
    assert j["process_group"]["world_size"] == 2 and j["process_group"]["backend"] == "gloo" and len(j["process_group"]["devices_by_rank"]) == 2
    assert j["collectives"]["all_to_all_ms"] > 0 and 0.0 < j["roofline"]["frac"] < 1.0
    assert abs(j["value"] - 2 * 1048576 / (j["ms_per_step"] * 1e-3)) < 1e-6 * j["value"]


def test_more_than_one_rank_prints_the_headline_before_any_side_configuration():
    """At N > 1 the stdout line goes out before the side configurations start (here: all of them cut by the budget), so a
    failure in one of them cannot cost the run its headline; the side file is rewritten afterwards."""
    j, err = _run(["--gpus", "2", "--rehearse-on-one-gpu", "--with-configs", "--steps", "4", "--warmup", "1", "--budget-seconds", "1"])
    assert j["n_gpus"] == 2 and j["config"]["index"] == "dealt" and "configs_skipped" not in j
    marks = [json.loads(ln[len("[bench-config] "):]) for ln in err.splitlines() if ln.startswith("[bench-config] ")]
    assert [m["name"] for m in marks] == ["headline"]
    side = json.loads((REPO / j["configs_file"]).read_text())
    assert len(side["configs_skipped"]) == 4 and side["configs"] == []
