"""CPU: host-side decisions of the layer stack that need no device -- which SA module takes the narrow-first-layer form
(csrc/narrow.hip), which layers run on padded copies, what the fused kernels accept."""
import numpy as np
import pytest
import torch


def test_narrow_first_layer_eligibility():
    from votenet_amd import mlp as M
    from votenet_amd import pointnet2 as P
    dev = torch.device("cpu")
    s = P.ParamStore(dev)
    sa1 = P.SAModule(s, "sa1", 2048, 0.2, 64, 3, [64, 64, 128], leaf=True)      # VoteNet's sa1: 6 grouped channels, no input gradient
    sa2 = P.SAModule(s, "sa2", 1024, 0.4, 64, 128, [128, 128, 256])            # 131 grouped channels
    two = P.SAModule(s, "two", 2048, 0.2, 64, 3, [64, 128], leaf=True)           # second layer is the pooled one: no room for the form
    inner = P.SAModule(s, "in", 2048, 0.2, 64, 3, [64, 64, 128])                 # same shapes, but its input takes a gradient
    rows = 8 * 2048 * 64
    assert sa1.narrow(rows) and not sa2.narrow(rows) and not two.narrow(rows) and not inner.narrow(rows)
    assert not sa1.narrow(rows + 1)  # whole 128-row tiles only
    old = P.NARROW_FIRST
    P.NARROW_FIRST = False
    try:
        assert not sa1.narrow(rows)
    finally:
        P.NARROW_FIRST = old
    assert M.narrow_supported(1024, 3, 64, 64) and M.narrow_supported(1024, 8, 128, 320)
    assert not M.narrow_supported(1024, 2, 64, 64) and not M.narrow_supported(1024, 6, 256, 64) and not M.narrow_supported(1024, 6, 64, 96)


def test_ragged_plain_layers_get_padded_copies():
    from votenet_amd import pointnet2 as P
    s = P.ParamStore(torch.device("cpu"))
    voting = P.make_mlp(s, "voting", 259, [256, 256, 259], "fc", last_plain=True)
    post = P.make_mlp(s, "post", 128, [128, 128, 79], "conv_post_", last_plain=True)
    assert [L.cout_pad for L in voting] == [0, 0, 320] and [L.cout_pad for L in post] == [0, 0, 128]
    assert [(L.bn, L.relu) for L in voting] == [(True, True), (True, True), (False, False)]


def test_fused_kernel_shape_predicates():
    from votenet_amd import mlp as M
    assert M.dgrad_bn_supported(1024, 128, 320) and M.dgrad_bn_supported(128, 512, 64)
    assert not M.dgrad_bn_supported(1000, 128, 128) and not M.dgrad_bn_supported(1024, 48, 128) and not M.dgrad_bn_supported(1024, 128, 96)
    assert M.linear_pool_supported(8192, 128, 256, 64) and not M.linear_pool_supported(8192, 128, 256, 32)
    assert M.group_linear_backward_supported(128, 64) and not M.group_linear_backward_supported(96, 64)


def test_narrow_statistics_algebra():
    """What votenet_narrow_stats / votenet_narrow_wgrad_first compute, in numpy: BatchNorm sums of z0 = u W0 + b0 and the first
    layer's weight gradient follow from the moments of u alone (the identity the never-stored layer rests on)."""
    rng = np.random.default_rng(3)
    n, k0, c0 = 5000, 6, 16
    u = rng.normal(size=(n, k0)) + np.array([0, 0, 0, 2.0, -1.0, 0.5])
    w0, b0 = rng.normal(size=(k0, c0)), rng.normal(size=c0)
    z0 = u @ w0 + b0
    m, M2 = u.sum(0), u.T @ u
    s1 = m @ w0 + n * b0
    s2 = np.einsum("dc,de,ec->c", w0, M2, w0) + 2 * b0 * (m @ w0) + n * b0 * b0
    assert np.allclose(s1, z0.sum(0), rtol=1e-12) and np.allclose(s2, (z0 * z0).sum(0), rtol=1e-12)
    A, B, C = rng.normal(size=c0), rng.normal(size=c0), rng.normal(size=c0)
    g = rng.normal(size=(n, c0)) * (rng.random(size=(n, c0)) > 0.4)
    dz0 = A * g + B + C * z0
    dw0 = A * (u.T @ g) + np.outer(m, B) + C * (M2 @ w0 + np.outer(m, b0))
    assert np.allclose(dw0, u.T @ dz0, rtol=1e-10, atol=1e-9)


def test_split_image_eligibility_and_cpu_store_is_inert():
    """Which weight matrices get a bf16 x 3 image (mlp.SplitImages), and that a store on the CPU (the gloo tests) never touches
    the device library for it."""
    from votenet_amd import mlp as M
    from votenet_amd import pointnet2 as P
    assert M.split_eligible(64, 64) and M.split_eligible(128, 256) and M.split_eligible(512, 256) and M.split_eligible(256, 320)
    assert not M.split_eligible(3, 64)       # first layers: the bounds-checked kernels
    assert not M.split_eligible(259, 256)    # ragged input width: its padded copy (320) is eligible instead
    assert not M.split_eligible(128, 79)     # ragged output width
    assert not M.split_eligible(1024, 256)   # beyond the fused kernel's cin
    imgs = M.SplitImages([torch.zeros(3, 64), torch.zeros(7)])  # nothing eligible: no buffer, no registration, refresh is a no-op
    assert imgs.nseg == 0
    imgs.refresh()
    imgs.close()
    s = P.ParamStore(torch.device("cpu"))
    P.SAModule(s, "sa", 64, 0.4, 16, 128, [128, 128, 256])
    s.materialize(0)
    s.enable_split(True)
    s.refresh_split()  # parameters not on a GPU: returns before any library call
    assert s._split_flat is None


def test_frozen_bn_table_is_thread_local_and_nests():
    """pointnet2.frozen_bn: the inference-mode BatchNorm table is set per thread and restored on exit (a predict() inside another
    model's pass, two models on two threads)."""
    import threading
    from votenet_amd import pointnet2 as P
    assert P._FROZEN.table is None
    seen = {}
    with P.frozen_bn({"a": 1}) as t:
        assert P._FROZEN.table is t
        with P.frozen_bn(None):
            assert P._FROZEN.table is None
        assert P._FROZEN.table == {"a": 1}

        def other():
            seen["other"] = P._FROZEN.table
            with P.frozen_bn({"b": 2}):
                seen["inner"] = P._FROZEN.table
        th = threading.Thread(target=other)
        th.start()
        th.join()
        assert P._FROZEN.table == {"a": 1}
    assert P._FROZEN.table is None and seen == {"other": None, "inner": {"b": 2}}


def test_param_store_generations():
    """Derived copies of the parameters belong to a generation of the bucket: params_changed() (the optimizer, a manual edit) opens a
    new one; invalidate_transposes() is the same call."""
    import torch
    from votenet_amd import pointnet2 as P
    st = P.ParamStore(torch.device("cpu"))
    P.make_mlp(st, "m", 32, [64], "fc")
    st.materialize(0)
    g0 = st.generation
    st.params_changed()
    st.invalidate_transposes()
    assert st.generation == g0 + 2 and st.t_event is None
    st.ensure_split()  # no images enabled on the CPU: a no-op, no library call
    assert st._split_gen != st.generation


def test_a_record_of_an_overwritten_batchnorm_block_is_refused():
    """Every BatchNorm layer's (scale | shift | mean | var) block is persistent (ParamStore.bn_flat): a record of forward pass k must not
    be read after pass k+1 rewrote the block (round-3 advice: gradients were silently wrong).  Host logic only."""
    import torch
    from votenet_amd import VotenetError
    from votenet_amd import pointnet2 as P
    store = P.ParamStore(torch.device("cpu"))
    layers = P.make_mlp(store, "m", 8, [8, 8], "fc")
    store.materialize(0)
    L = layers[0]
    assert L.bn and L.name in store._bn_views
    rec1 = dict(layer=L, bn_pass=store.bn_block_written(L.name))
    P.check_bn_block(rec1)
    P.check_bn_block(dict(layer=L))  # a record with its own buffer (frozen BatchNorm, no persistent block): nothing to check
    rec2 = dict(layer=L, bn_pass=store.bn_block_written(L.name))
    P.check_bn_block(rec2)
    with pytest.raises(VotenetError, match="rewritten by a later training-mode pass"):
        P.check_bn_block(rec1)
