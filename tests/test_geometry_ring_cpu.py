"""CPU: the bookkeeping around model.GeometryGraph (which graph of the ring serves the next prefetch, when the chain falls back to
launches, which hand-outs are trusted) with a stand-in for the graph itself -- the captures and replays proper are GPU tests
(tests/test_gpu_model.py::test_geometry_graph_replays_are_the_launch_by_launch_chain)."""
import types

import pytest
import torch

from votenet_amd import mlp as M
from votenet_amd import model as VM
from votenet_amd import pointnet2 as P


class _Graph:
    made = 0

    def __init__(self, net, x, side):
        _Graph.made += 1
        self.generation = 0


@pytest.fixture
def net(monkeypatch):
    monkeypatch.setattr(VM, "GeometryGraph", _Graph)
    monkeypatch.setattr(VM, "GEOMETRY_GRAPHS", True)
    monkeypatch.setattr(torch.cuda, "is_current_stream_capturing", lambda: False)
    _Graph.made = 0
    n = types.SimpleNamespace(proposal=types.SimpleNamespace(npoint=256))
    n.pick = lambda x: VM.VoteNetHotPath._geometry_graph(n, x, None)
    return n


def test_ring_order_and_warm_up(net):
    x = torch.zeros(2, 64, 3)
    assert net.pick(x) is None                      # the first chain of a shape runs launch by launch (sizes the library's scratch)
    a, b, c = net.pick(x), net.pick(x), net.pick(x)  # then one capture per ring slot
    assert _Graph.made == VM.GEOMETRY_RING == 3 and len({id(a), id(b), id(c)}) == 3
    assert [net.pick(x) for _ in range(6)] == [a, b, c, a, b, c]  # least recently used first
    assert _Graph.made == 3


def test_the_graph_under_the_running_pass_is_skipped(net):
    x = torch.zeros(2, 64, 3)
    net.pick(x)
    a, b, c = net.pick(x), net.pick(x), net.pick(x)
    net._geometry_current = a
    assert net.pick(x) is None and net.pick(x) is b and net.pick(x) is c  # a's turn passes: that prefetch runs launch by launch
    net._geometry_current = None
    assert net.pick(x) is a


def test_fallbacks_and_configuration_changes(net, monkeypatch):
    x, y = torch.zeros(2, 64, 3), torch.zeros(2, 128, 3)
    net.pick(x)
    a = net.pick(x)
    monkeypatch.setattr(VM, "GEOMETRY_GRAPHS", False)
    assert net.pick(x) is None
    monkeypatch.setattr(VM, "GEOMETRY_GRAPHS", True)
    monkeypatch.setattr(M, "DETERMINISTIC", True)
    assert net.pick(x) is None                       # inverse indices ride on tensors as attributes: launches
    monkeypatch.setattr(M, "DETERMINISTIC", False)
    monkeypatch.setattr(P.tf_sampling, "PROFILE_EVENTS", [])
    assert net.pick(x) is None                       # per-launch events wanted
    monkeypatch.setattr(P.tf_sampling, "PROFILE_EVENTS", None)
    assert net.pick(y) is None and net.pick(y) is not a          # another shape: its own warm-up and ring
    assert len(net._geometry_rings) == 2
    monkeypatch.setattr(P, "HALF_GROUPS", not P.HALF_GROUPS)     # a third configuration: the old rings (and their buffers) go
    assert net.pick(x) is None and len(net._geometry_rings) == 1


def test_stale_hand_outs_are_not_trusted(net):
    x = torch.zeros(2, 64, 3)
    g = _Graph(None, x, None)
    g.generation = 5
    net._prefetched = {id(x): (x, x._version, {"sa1": 1}, {"sa1": 2}, g, 5)}
    assert VM.VoteNetHotPath._take_prefetched(net, x) == ({"sa1": 1}, {"sa1": 2}) and net._geometry_current is g
    net._prefetched = {id(x): (x, x._version, {"sa1": 1}, {"sa1": 2}, g, 4)}  # the graph has served another batch since
    assert VM.VoteNetHotPath._take_prefetched(net, x) is None and net._geometry_current is None
    net._prefetched = {id(x): (x, x._version - 1 if x._version else -1, {}, {}, None, 0)}  # the tensor was written since
    assert VM.VoteNetHotPath._take_prefetched(net, x) is None


def test_changing_shapes_turn_the_graphs_off(net):
    with pytest.warns(UserWarning, match="launch by launch"):
        for i in range(4 + 2 * VM.GEOMETRY_MAX_EVICTIONS):  # two shapes are kept side by side: every second new one evicts
            x = torch.zeros(1, 32 + i, 3)
            net.pick(x)
            net.pick(x)
    assert net._geometry_graphs_off and net.pick(torch.zeros(1, 32, 3)) is None
    made = _Graph.made
    net.pick(torch.zeros(1, 32, 3))
    assert _Graph.made == made


def test_a_tape_whose_geometry_was_overwritten_is_refused():
    """forward() stamps the tape with (graph, generation) when its geometry lives in a GeometryGraph's buffers; backward() refuses the
    tape once that graph has replayed for another batch (round-3 advice: it used to differentiate through the other batch's lists)."""
    from votenet_amd import VotenetError
    g = _Graph(None, None, None)
    g.generation = 3
    tape = [dict(op="sa", geometry_stamp=(g, 3)), dict(op="sa")]
    VM.VoteNetHotPath.check_tape(tape)              # same generation: fine
    VM.VoteNetHotPath.check_tape([dict(op="sa")])   # geometry computed in place (fresh tensors): no stamp, nothing to check
    VM.VoteNetHotPath.check_tape([])
    g.generation = 4
    with pytest.raises(VotenetError, match="overwritten by 1 later prefetch"):
        VM.VoteNetHotPath.check_tape(tape)


def test_private_arena_borrows_and_restores_the_pass_arena():
    """model._PrivateArena (the scratch of a captured stretch): requests made inside come out of the private buffer, the enclosing pass's
    arena state -- cursors included -- is back afterwards, and the demand the stretch made is reported."""
    a = M._StatsArena
    saved = {k: getattr(a, k) for k in VM._PrivateArena.FIELDS}
    try:
        a.buf, a.nd, a.cap32, a.zeroed32 = torch.zeros(64 + 128, dtype=torch.float64), 64, 256, 256
        a.off, a.off32, a.want32, a.depth, a.active, a.whole_step = 10, 20, 24, 1, True, False
        outer = a.buf
        priv = torch.zeros(32 + 64, dtype=torch.float64)
        with VM._PrivateArena(priv, 32) as pa:
            assert a.buf is priv and a.off == 0 and a.off32 == 0 and a.cap32 == 128 and a.whole_step
            v64 = M._zeros_f64(6, torch.device("cpu"))
            v32 = M._zeros_f32((5,), torch.device("cpu"))
            assert v64.data_ptr() == priv.data_ptr() and v32.data_ptr() == priv.data_ptr() + 32 * 8
        assert pa.used == (6, 8)  # doubles, floats (rounded up to 16 bytes)
        assert a.buf is outer and (a.off, a.off32, a.want32, a.whole_step) == (10, 20, 24, False)
    finally:
        for k, v in saved.items():
            setattr(a, k, v)


def test_outputs_of_a_replayed_step_refuse_stale_reads():
    from votenet_amd import VotenetError
    g = types.SimpleNamespace(replays=3)
    out = VM._StretchOutputs({"votes_xyz": torch.zeros(2)}, g)
    assert out["votes_xyz"].shape == (2,) and "votes_xyz" in out
    g.replays = 4  # the capture ran again: its pool holds a later step's values
    with pytest.raises(VotenetError, match="replayed for a later step"):
        out["votes_xyz"]
