"""CPU: the pieces of bench.py that run without a GPU -- the cpu_baseline leg (the oracle timed on one scene) and the
argument parser -- so that a broken default run is caught here and not on the GPU box."""
import sys

import bench


def test_cpu_baseline_leg_runs_and_reports_the_contract_fields():
    out = bench.cpu_baseline(2560, "room", min_seconds=0.0)
    assert set(out) == {"value", "unit", "cores", "kind", "sample"}
    assert out["unit"] == "scenes/s" and out["cores"] == 1 and out["kind"] == "port" and out["value"] > 0


def test_default_arguments_are_one_gpu_and_a_short_run(monkeypatch):
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = bench.parse()
    assert a.gpus == 1 and a.steps <= 50 and a.batch == 8 and a.points == 20480 and not a.no_pipeline
