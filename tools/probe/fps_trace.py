"""Per-phase cycle accounting of one FPS round (wave 0 of scene 0), from an instrumented build of fps.hip (-DFPS_TRACE):
    hipcc ... -DFPS_TRACE -shared votenet_amd/csrc/fps.hip votenet_amd/csrc/common.hip -o tools/probe/lib/libfps_trace.so"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from votenet_amd import synth
L = ctypes.CDLL(os.path.join(ROOT, "tools", "probe", "lib", "libfps_trace.so"))
L.votenet_fps_temp_floats.restype = ctypes.c_size_t
L.votenet_fps_temp_floats.argtypes = [ctypes.c_int, ctypes.c_int]
L.votenet_farthest_point_sample.argtypes = [ctypes.c_int] * 3 + [ctypes.c_void_p] * 4
L.votenet_fps_debug_two_pick.restype = None
L.votenet_fps_debug_two_pick(int(os.environ.get("TWO_PICK", "0")))
dev = torch.device("cuda:0")
for kind in ("room", "uniform"):
    b, n, m = 8, 20480, 2048
    x = torch.from_numpy(synth.room_batch(b, n, 1000) if kind == "room" else synth.uniform_batch(b, n, 1000)).to(dev)
    temp = torch.empty(L.votenet_fps_temp_floats(b, n), dtype=torch.float32, device=dev)
    out = torch.empty((b, m), dtype=torch.int32, device=dev)
    buf = (ctypes.c_ulonglong * 8)()
    L.votenet_farthest_point_sample(b, n, m, x.data_ptr(), temp.data_ptr(), out.data_ptr(), None)
    L.votenet_fps_trace_read(buf, 1)
    L.votenet_farthest_point_sample(b, n, m, x.data_ptr(), temp.data_ptr(), out.data_ptr(), None)
    L.votenet_fps_trace_read(buf, 1)
    names = ["loop/out", "box tests", "touched buckets", "wave winner", "cross-wave", "stage 2 (two-pick kernel)"]
    tot = sum(buf[:6])
    print(kind, "cycles per round (s_memtime ticks, 100 MHz = 10 ns each?):", round(tot / (m - 1), 1))
    for i, nm in enumerate(names):
        print("   %-16s %8.1f per round  %5.1f%%" % (nm, buf[i] / (m - 1), 100.0 * buf[i] / tot))
