"""pytest configuration: the `gpu` marker, import paths, and the shared oracle / library fixtures."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The reference's device kernels (oracle/_ref/libref_*_gpu.so) travel with the snapshot.  When the directory is there the
    # three-way tests of test_gpu_reference_kernels.py must RUN: a library that then fails to load is a failure, not a skip.
    if os.path.isdir(os.path.join(ROOT, "oracle", "_ref")) and "VOTENET_REQUIRE_REF" not in os.environ:
        os.environ["VOTENET_REQUIRE_REF"] = "1"


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
def linklib(hiplib):
    """tests/link/launcher_link.cpp (the eight launcher declarations of tf_sampling.cpp:65,94,125,150 / tf_grouping.cpp:66,108,142,173)
    linked against the product with -Wl,-z,defs: an unexported or mis-typed launcher is an undefined symbol and the link fails, as
    dlopen(RTLD_NOW) of the reference's real wrapper would.  Returns the ctypes handle of the linked object."""
    import ctypes
    import subprocess
    from votenet_amd import _lib
    libdir = os.path.dirname(_lib.lib_path())
    out_dir = os.path.join(ROOT, "tests", "link", "_build")
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, "liblauncher_link.so")
    cmd = ["g++", "-std=c++11", "-shared", "-fPIC", "-O2", "-Wl,-z,defs", os.path.join(ROOT, "tests", "link", "launcher_link.cpp"),
           "-o", out, "-L" + libdir, "-lvotenet_hip", "-Wl,-rpath," + libdir]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, "the drop-in link line fails:\n%s\n%s" % (" ".join(cmd), r.stderr)
    # RTLD_LOCAL: the product's launcher names must not enter the process's global symbol scope (the reference libraries of
    # oracle/_ref define the same names; they are linked -Bsymbolic as well)
    return ctypes.CDLL(out, mode=ctypes.RTLD_LOCAL | os.RTLD_NOW)


@pytest.fixture(scope="session")
def dev():
    import torch
    assert torch.cuda.is_available(), "gpu-marked test run without a GPU"
    return torch.device("cuda:0")


@pytest.fixture(params=[2, 1, 0], ids=["fp16x2_forward_images", "bf16x3_images", "fp32_mfma"])
def gemm_form(request, hiplib):
    """The fused GEMMs on split-operand images of the weights -- 2: the model's default (forward matrices as fp16 x 2, the transposed
    copies of the backward pass as bf16 x 3: mlp.FORWARD_H2); 1: bf16 x 3 everywhere (the round-3..5 form) -- or, 0, on the fp32 MFMA
    kernels (votenet_debug_fast_bf3(0): registered images are ignored).  A test that wants images registers its matrices with
    mlp.SplitImages; a ParamStore picks the form when it first builds its images, i.e. after this fixture ran."""
    from votenet_amd import mlp as M
    prev_h2 = M.FORWARD_H2
    M.FORWARD_H2 = request.param == 2
    on = 1 if request.param else 0
    for name in ("fast_bf3", "gram_bf3", "wgrad_bf3"):  # through mlp.debug_switch: graphs captured under the other form are not reused
        M.debug_switch(name, 3 if (name == "gram_bf3" and request.param == 1) else on)
    yield request.param
    M.FORWARD_H2 = prev_h2
    for name in ("fast_bf3", "gram_bf3", "wgrad_bf3"):
        M.debug_switch(name, 1)
