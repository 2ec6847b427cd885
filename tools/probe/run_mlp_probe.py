import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from votenet_amd import _lib
dev = torch.device("cuda:0")
rows, ci, co = 524288, 256, 128
x = torch.randn(rows, ci, device=dev); w = torch.randn(ci, co, device=dev); z = torch.empty(rows, co, device=dev)
sc = torch.ones(ci, device=dev); sh = torch.zeros(ci, device=dev); stats = torch.zeros(2 * co, dtype=torch.float64, device=dev)
here = os.path.dirname(os.path.abspath(__file__))
for v in ["BASE", "L2ONLY", "L2STORE", "L2ONLY_L2STORE", "NOLOAD", "NOLOAD_NOEPI"]:
    L = ctypes.CDLL(os.path.join(here, "lib", "libmlp_%s.so" % v))
    fn = L.votenet_mlp_linear
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.POINTER(_lib.MlpInput), ctypes.c_long, ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 5
    for name, affine in (("dgrad-like", False), ("fwd-like", True)):
        d = _lib.MlpInput(); d.x = x.data_ptr()
        if affine:
            d.in_scale, d.in_shift, d.in_relu = sc.data_ptr(), sh.data_ptr(), 1
        st = stats.data_ptr() if affine else None
        def run():
            fn(ctypes.byref(d), rows, ci, co, w.data_ptr(), None, z.data_ptr(), st, None)
        for _ in range(3): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(torch.cuda.default_stream())
        for _ in range(10): run()
        e1.record(torch.cuda.default_stream()); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 10
        print("%-10s %-10s %.3f ms  %.1f TF" % (v, name, t, 2.0 * rows * ci * co / t / 1e9))
