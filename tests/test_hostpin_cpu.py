"""CPU: votenet_amd/hostpin.py against a made-up sysfs (two NUMA nodes, four GPUs) -- which CPUs a rank is confined to."""
import os

import pytest

from votenet_amd import hostpin


def test_cpulist():
    assert hostpin._cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11] and hostpin._cpulist("") == []


@pytest.fixture
def fake(tmp_path, monkeypatch):
    kfd, pci = tmp_path / "kfd", tmp_path / "pci"
    for i, (simd, loc) in enumerate([(0, 0), (0, 0), (1024, 0x0500), (1024, 0x1500), (1024, 0x8500), (1024, 0x9500)]):
        d = kfd / str(i)
        d.mkdir(parents=True)
        (d / "properties").write_text("cpu_cores_count %d\nsimd_count %d\nlocation_id %d\ndomain 0\n" % (64 if simd == 0 else 0, simd, loc))
    for bus, node in ((0x05, 0), (0x15, 0), (0x85, 1), (0x95, 1)):
        d = pci / ("0000:%02x:00.0" % bus)
        d.mkdir(parents=True)
        (d / "numa_node").write_text("%d\n" % node)
    state = {"mask": set(range(256))}
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(state["mask"]), raising=False)
    monkeypatch.setattr(os, "sched_setaffinity", lambda pid, cpus: state.__setitem__("mask", set(cpus)), raising=False)
    real_nodes = hostpin.gpu_numa_nodes
    monkeypatch.setattr(hostpin, "gpu_numa_nodes", lambda: real_nodes(str(kfd), str(pci)))
    real_open = open

    def fake_open(path, *a, **k):
        if isinstance(path, str) and path.startswith("/sys/devices/system/node/node"):
            n = int(path.split("node")[-1].split("/")[0])
            import io
            return io.StringIO("0-63,128-191\n" if n == 0 else "64-127,192-255\n")
        return real_open(path, *a, **k)
    monkeypatch.setattr("builtins.open", fake_open)
    for v in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "VOTENET_NO_PIN"):
        monkeypatch.delenv(v, raising=False)
    return state


def test_numa_nodes_in_kfd_order(fake):
    assert hostpin.gpu_numa_nodes() == [0, 0, 1, 1]


def test_ranks_take_blocks_of_their_gpus_node(fake):
    assert hostpin.pin(0) == list(range(8, 16))       # block 1 of node 0 (block 0 holds CPU 0)
    fake["mask"] = set(range(256))
    assert hostpin.pin(1) == list(range(16, 24))      # the second GPU of node 0: the next block
    fake["mask"] = set(range(256))
    assert hostpin.pin(2) == list(range(72, 80))      # node 1
    fake["mask"] = set(range(256))
    assert hostpin.pin(3) == list(range(80, 88))
    assert fake["mask"] == set(range(80, 88))


def test_no_ops(fake, monkeypatch):
    fake["mask"] = set(range(12))
    assert hostpin.pin(0) is None and fake["mask"] == set(range(12))   # already narrow: a launcher chose
    fake["mask"] = set(range(256))
    monkeypatch.setenv("VOTENET_NO_PIN", "1")
    assert hostpin.pin(0) is None
    monkeypatch.delenv("VOTENET_NO_PIN")
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "3")
    assert hostpin.pin(0) == list(range(80, 88))      # device 0 of this process is physical GPU 3


def test_visible_devices_mapping(monkeypatch):
    """ROCR_VISIBLE_DEVICES filters what the runtime sees, HIP_VISIBLE_DEVICES indexes into that list; an entry that is not an index
    (a uuid) leaves the rank unpinned with a warning instead of pinning it to some other GPU's node (round-3 advice)."""
    import pytest
    P = hostpin.physical_gpu
    assert P(2, 8, {}) == 2 and P(8, 8, {}) is None
    assert P(0, 8, {"HIP_VISIBLE_DEVICES": "3,5"}) == 3 and P(1, 8, {"HIP_VISIBLE_DEVICES": "3,5"}) == 5
    assert P(1, 8, {"ROCR_VISIBLE_DEVICES": "4,5,6,7", "HIP_VISIBLE_DEVICES": "2,0"}) == 4   # HIP's index 0 = ROCR's entry 0 = GPU 4
    assert P(0, 8, {"ROCR_VISIBLE_DEVICES": "4,5,6,7", "HIP_VISIBLE_DEVICES": "2,0"}) == 6
    assert P(0, 8, {"CUDA_VISIBLE_DEVICES": "7"}) == 7
    assert P(0, 8, {"CUDA_VISIBLE_DEVICES": "7", "HIP_VISIBLE_DEVICES": "1"}) == 1            # HIP_VISIBLE_DEVICES wins
    assert P(1, 8, {"HIP_VISIBLE_DEVICES": "1,9,2"}) is None                                  # the list ends at the first bad index
    with pytest.warns(UserWarning, match="left unpinned"):
        assert P(0, 8, {"HIP_VISIBLE_DEVICES": "GPU-abcdef0123456789"}) is None


def test_uuid_entries_leave_the_rank_unpinned(fake, monkeypatch):
    import pytest
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "GPU-0123456789abcdef")
    with pytest.warns(UserWarning):
        assert hostpin.pin(0) is None
    assert fake["mask"] == set(range(256))


def test_bench_loads_hostpin_without_the_package():
    """bench.py pins before torch exists in the process: the module is loaded by path, not through votenet_amd/__init__ (which imports
    torch).  A fresh interpreter proves it."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); import bench; h = bench.load_hostpin(); "
            "assert 'torch' not in sys.modules and 'votenet_amd' not in sys.modules; assert callable(h.pin) and callable(h.unpin); print('ok')" % root)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stderr


def test_eight_gpus_two_nodes_every_rank_gets_its_own_block(tmp_path, monkeypatch):
    """The node of BASELINE config 4 (8 GPUs, 4 per NUMA node, 2 x 64 cores + SMT): eight ranks -> eight disjoint blocks of eight CPUs,
    each on its GPU's node, none holding CPU 0."""
    kfd, pci = tmp_path / "kfd", tmp_path / "pci"
    buses = [0x05, 0x15, 0x25, 0x35, 0x85, 0x95, 0xa5, 0xb5]
    for i in range(2):
        d = kfd / str(i)
        d.mkdir(parents=True)
        (d / "properties").write_text("cpu_cores_count 64\nsimd_count 0\nlocation_id 0\ndomain 0\n")
    for i, bus in enumerate(buses):
        d = kfd / str(2 + i)
        d.mkdir(parents=True)
        (d / "properties").write_text("cpu_cores_count 0\nsimd_count 1024\nlocation_id %d\ndomain 0\n" % (bus << 8))
        dp_ = pci / ("0000:%02x:00.0" % bus)
        dp_.mkdir(parents=True)
        (dp_ / "numa_node").write_text("%d\n" % (0 if i < 4 else 1))
    state = {"mask": set(range(256))}
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(state["mask"]), raising=False)
    monkeypatch.setattr(os, "sched_setaffinity", lambda pid, cpus: state.__setitem__("mask", set(cpus)), raising=False)
    real_nodes = hostpin.gpu_numa_nodes
    monkeypatch.setattr(hostpin, "gpu_numa_nodes", lambda: real_nodes(str(kfd), str(pci)))
    real_open = open

    def fake_open(path, *a, **k):
        if isinstance(path, str) and path.startswith("/sys/devices/system/node/node"):
            import io
            n = int(path.split("node")[-1].split("/")[0])
            return io.StringIO("0-63,128-191\n" if n == 0 else "64-127,192-255\n")
        return real_open(path, *a, **k)
    monkeypatch.setattr("builtins.open", fake_open)
    for v in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "VOTENET_NO_PIN"):
        monkeypatch.delenv(v, raising=False)
    assert hostpin.gpu_numa_nodes() == [0, 0, 0, 0, 1, 1, 1, 1]
    blocks = []
    for rank in range(8):
        state["mask"] = set(range(256))
        got = hostpin.pin(rank)
        assert got is not None and len(got) == 8 and 0 not in got
        node0 = set(range(0, 64)) | set(range(128, 192))
        assert set(got) <= (node0 if rank < 4 else set(range(256)) - node0), (rank, got)
        blocks.append(set(got))
    assert len(set().union(*blocks)) == 64          # pairwise disjoint
