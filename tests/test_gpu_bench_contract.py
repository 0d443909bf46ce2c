"""bench.py contract (the driver parses this line) on a small workload, through the real pipelined loop."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.spawns]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_contract_line():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "C1", "--steps", "6", "--warmup", "2",
                        "--cpu-frames", "1", "--cpu-reps", "3"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "cpu_baseline_sweep", "self_check",
              "oracle_check", "planted_top1"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2 and d["value"] > 0 and d["unit"] == "faces/s"
    assert d["scaling"] == "weak" and d["higher_is_better"] is True and d["vs_baseline"] is None
    rf = d["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1 and "workload" in d["config"]
    assert [r["cores"] for r in d["cpu_baseline_sweep"]][0] == 1 and cb["host_cpu_count"] >= cb["cores"]
    assert cb["value"] == max(r["value"] for r in d["cpu_baseline_sweep"])
    oc = d["oracle_check"]                  # the CPU oracle's ids / boxes / embeddings for batch 0's frames == the GPU's
    assert oc["ok"] and oc["ids_equal"] and oc["faces"] >= 1 and oc["min_cos"] >= 1 - 1e-3
    assert d["planted_top1"]["faces"] > 0
