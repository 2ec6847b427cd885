"""pytest configuration: the `gpu` marker, import paths, and the shared oracle / library fixtures."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def O():
    """The CPU oracle (checker).  Built on demand."""
    from oracle import oracle as orc
    orc.lib()
    return orc


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"), allow_pickle=False)
    return load


@pytest.fixture(scope="session")
def hiplib():
    """The product library; built on demand (hipcc cross-compiles without a GPU)."""
    import votenet_amd
    from votenet_amd import _lib
    if not os.path.exists(_lib.lib_path()):
        votenet_amd.build()
    return _lib.lib()


@pytest.fixture(scope="session")
def dev():
    import torch
    assert torch.cuda.is_available(), "gpu-marked test run without a GPU"
    return torch.device("cuda:0")


@pytest.fixture(params=[1, 0], ids=["bf16x3_images", "fp32_mfma"])
def gemm_form(request, hiplib):
    """The fused GEMMs on bf16 x 3 images of the weights (the model's default; a test that wants them registers its matrices with
    mlp.SplitImages) or on the fp32 MFMA kernels (votenet_debug_fast_bf3(0): registered images are ignored)."""
    from votenet_amd import mlp as M
    for name in ("fast_bf3", "gram_bf3", "wgrad_bf3"):  # through mlp.debug_switch: graphs captured under the other form are not reused
        M.debug_switch(name, request.param)
    yield request.param
    for name in ("fast_bf3", "gram_bf3", "wgrad_bf3"):
        M.debug_switch(name, 1)
