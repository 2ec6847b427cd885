"""CPU: the pieces of bench.py that run without a GPU -- the cpu_baseline leg (the oracle timed on one scene) and the
argument parser -- so that a broken default run is caught here and not on the GPU box."""
import sys

import bench


def test_cpu_baseline_leg_runs_and_reports_the_contract_fields():
    out = bench.cpu_baseline(2560, "room", min_seconds=0.0, max_scenes=1, ops=False)
    assert {"value", "unit", "cores", "kind", "sample", "value_1t", "value_all", "cpu_model", "physical_cores"} <= set(out)
    assert out["unit"] == "scenes/s" and out["cores"] >= 1 and out["kind"] == "port" and out["value_1t"] > 0 and out["value_all"] > 0
    assert out["value"] == out["value_all"] and isinstance(out["cpu_model"], str)


def test_ball_query_scanned_pair_count_from_the_outputs():
    """bench_legs.ball_query_pairs: the reference stops a query at its K-th hit; the full-scan kernel stops a 64-query workgroup
    when all are full, in 4096-candidate super-chunks."""
    import os
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import bench_legs
    n, k = 10000, 4
    idx = torch.zeros(1, 128, k, dtype=torch.int32)
    cnt = torch.full((1, 128), k, dtype=torch.int32)
    idx[0, :, k - 1] = 99                      # every query of both workgroups full after candidate 99
    idx[0, 70, k - 1] = 5000                   # ... except one query of the second workgroup: needs two super-chunks
    cnt[0, 3] = 2                              # a query of the first workgroup never fills: full scan for its workgroup
    p = bench_legs.ball_query_pairs(idx, cnt, n)
    assert p["all_pairs"] == 128 * n
    assert p["reference_algorithm"] == 126 * 100 + 5001 + n
    assert p["kernel"] == 64 * n + 64 * 8192


def test_default_arguments_are_one_gpu_and_a_short_run(monkeypatch):
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = bench.parse()
    assert a.gpus == 1 and a.steps <= 50 and a.batch == 8 and a.points == 20480 and not a.no_pipeline


def _run_bench(*argv, env=None):
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(root, "bench.py")] + list(argv), env=e, capture_output=True, text=True, timeout=300)


def test_gpus_n_starts_n_ranks_itself():
    """`python bench.py --gpus 2` with no launcher around it: two child ranks rendezvous (gloo here, RCCL on the GPU box),
    all-reduce, and rank 0's line says n_gpus 2 (round-1 verdict: --gpus was parsed and ignored)."""
    import json
    r = _run_bench("--gpus", "2", "--dry-run")
    assert r.returncode == 0, r.stderr
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["rank_sum"] == 3.0
    # world > 1: the line validates its own data-parallel path (round-3 verdict item 5) -- the keys a real `--gpus N` line carries,
    # produced here by the same bench.dp_self_check over gloo on stand-in replicas
    c = line["check_dp"]
    assert c["equal_everywhere"] and c["equal_on_this_rank"] and c["ranks_identical"]
    assert c["world_size"] == 2 and c["distinct_devices"] == 2 and c["backend"] == "gloo" and c["steps"] == 2
    assert [w for w, _ in c["collectives_per_step"]] == ["tail", "head"]
    assert "tail_exposed_ms" in c
    # ... and every rank's own clock / host placement (a straggler is visible in the one SCALE record)
    pr = line["per_rank"]
    assert [r_["rank"] for r_ in pr["ranks"]] == [0, 1] and [r_["ms_per_step"] for r_ in pr["ranks"]] == [1.0, 2.0]
    assert pr["ms_per_step_min"] == 1.0 and pr["ms_per_step_max"] == 2.0
    assert all(set(r_["hostpin"]) == {"cpus", "gpu_numa_node"} for r_ in pr["ranks"])


def test_gpus_8_starts_eight_ranks_itself():
    """The first real `--gpus 8` run must not die on rank plumbing: eight ranks started by bench.py itself rendezvous (gloo here),
    every rank takes part in the self-check's collectives and reports its own clock / host pin, and rank 0 prints ONE line."""
    import json
    r = _run_bench("--gpus", "8", "--dry-run")
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["rank_sum"] == float(sum(range(1, 9)))
    c = line["check_dp"]
    assert c["equal_everywhere"] and c["ranks_identical"] and c["world_size"] == 8 and c["distinct_devices"] == 8
    pr = line["per_rank"]
    assert [r_["rank"] for r_ in pr["ranks"]] == list(range(8))
    assert pr["ms_per_step_min"] == 1.0 and pr["ms_per_step_max"] == 8.0


def test_diverged_replicas_fail_the_run():
    """bench.dp_self_check's verdict decides the exit code: replicas that differ after the check's steps -> exit code 4 on every rank."""
    r = _run_bench("--gpus", "2", "--dry-run", env={"VOTENET_BENCH_DRYRUN_DIVERGE": "1"})
    assert r.returncode != 0
    import json
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["check_dp"]["equal_everywhere"] is False


def test_a_failed_rank_fails_the_run():
    r = _run_bench("--gpus", "2", "--dry-run", env={"VOTENET_BENCH_DRYRUN_FAIL_RANK": "1"})
    assert r.returncode != 0 and "rank(s) failed" in r.stderr


def test_gpus_must_match_the_launcher_world_size():
    r = _run_bench("--gpus", "4", "--dry-run", env={"WORLD_SIZE": "2", "RANK": "0"})
    assert r.returncode != 0 and "does not match WORLD_SIZE" in r.stderr


def test_gpus_are_counted_in_sysfs_without_touching_the_runtime(tmp_path, monkeypatch):
    """The launcher parent must not initialise HIP (it fork+execs the ranks): GPUs = KFD topology nodes with simd_count > 0,
    narrowed by *_VISIBLE_DEVICES; no topology -> None (the ranks then check for themselves)."""
    for i, simd in enumerate([0, 0, 1024, 1024, 1024]):  # two CPU nodes, three GPUs
        d = tmp_path / str(i)
        d.mkdir()
        (d / "properties").write_text("cpu_cores_count %d\nsimd_count %d\nmem_banks_count 1\n" % (64 if simd == 0 else 0, simd))
    for v in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(v, raising=False)
    assert bench.visible_gpu_count(str(tmp_path)) == 3
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,2")
    assert bench.visible_gpu_count(str(tmp_path)) == 2
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    assert bench.visible_gpu_count(str(tmp_path / "missing")) is None


def test_the_launcher_parent_never_imports_torch():
    """launch_ranks runs before anything GPU-related is imported: the parent process of `--gpus N` holds no HIP runtime."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.argv = ['bench.py', '--gpus', '2', '--dry-run']; sys.path.insert(0, %r); import bench\n"
            "a = bench.parse(); rc = bench.launch_ranks(a)\n"
            "assert rc == 0, rc\n"
            "assert 'torch' not in sys.modules, 'the launcher imported torch'\n" % root)
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
