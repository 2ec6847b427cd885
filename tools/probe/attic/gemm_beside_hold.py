"""One forward GEMM (524288 x 128 -> 128) beside a kernel that only OCCUPIES CUs (tools/probe/src/hold.hip): 8 workgroups of 768
threads with 97 KB of LDS (the FPS kernel's footprint), asleep or spinning, against the real FPS kernel.
Build the holder first (hipcc cross-compiles without a GPU):
    hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/probe/src/hold.hip -o tools/probe/lib/libhold.so"""
import ctypes, os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [R]
import torch
from votenet_amd import mlp as M, synth, tf_sampling
hold = ctypes.CDLL(os.path.join(R, "tools", "probe", "lib", "libhold.so"))
hold.hold_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_long, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
dev = torch.device("cuda:0")
x0 = torch.from_numpy(synth.room_batch(8, 20480, 1)).to(dev)
side = torch.cuda.Stream(device=dev)
sink = torch.zeros(4, device=dev)
rows, c, co = 524288, 128, 128
x = torch.randn(rows, c, device=dev); w = torch.randn(c, co, device=dev) * 0.1
sc, sh = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev)
def gemms(n):
    for _ in range(n):
        M.linear_dense(x, w, None, sc, sh, True)
def beside(name, fn):
    gemms(3); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if fn is not None:
        with torch.cuda.stream(side):
            fn()
        torch.cuda._sleep(300000)
    e0.record(); gemms(6); e1.record(); torch.cuda.synchronize()
    print("%-64s %.4f ms per GEMM" % (name, e0.elapsed_time(e1) / 6), flush=True)
CY = 2200 * 4000  # ~4 ms of shader clocks at 2.2 GHz
def holder(grid, block, lds, spin):
    return lambda: hold.hold_launch(grid, block, lds, CY, spin, sink.data_ptr(), torch.cuda.current_stream().cuda_stream)
beside("alone", None)
beside("beside 3 x FPS sa1 (8 workgroups x 768 threads, 97 KB LDS)", lambda: [tf_sampling.farthest_point_sample(2048, x0) for _ in range(3)])
beside("beside 8 sleeping workgroups x 768 threads, 97 KB LDS", holder(8, 768, 98944, 0))
beside("beside 8 spinning workgroups x 768 threads, 97 KB LDS", holder(8, 768, 98944, 1))
beside("beside 8 sleeping workgroups x 768 threads, 160000 B LDS", holder(8, 768, 160000, 0))
beside("beside 8 sleeping workgroups x 64 threads, 1 KB LDS", holder(8, 64, 1024, 0))
beside("beside 64 sleeping workgroups x 768 threads, 97 KB LDS", holder(64, 768, 98944, 0))
# round 3: what if the FPS kernel's footprint were smaller (tie keys as 16-bit, 49 KB; or all in registers)?
beside("beside 8 sleeping workgroups x 768 threads, 49 KB LDS", holder(8, 768, 50176, 0))
beside("beside 8 spinning workgroups x 768 threads, 49 KB LDS", holder(8, 768, 50176, 1))
beside("beside 8 sleeping workgroups x 768 threads, 1 KB LDS", holder(8, 768, 1024, 0))
beside("beside 8 sleeping workgroups x 256 threads, 49 KB LDS", holder(8, 256, 50176, 0))
