"""Where the cycles of one sa1 FPS round go (fps_bucket_kernel<12,32>, 8 x 20480 -> 2048), two independent ways:
  (1) s_memtime stamps between the phases of wave 0 of scene 0 (instrumented build; each stamp waits for lgkmcnt(0), so
      the instrumented round is longer than the real one -- both totals are printed);
  (2) ablations of the UN-instrumented kernel: phases removed one after the other (the indices are then wrong; only the
      time is read): full kernel -> no touched-bucket work -> no box tests either -> no cross-wave exchange either.
Build first: tools/probe/fps_round_trace.sh (here, cross-compiled); run on the GPU box."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from votenet_amd import synth
dev = torch.device("cuda:0")
b, n, m = 8, 20480, 2048
CLK = 2.4e9


def load(name):
    L = ctypes.CDLL(os.path.join(ROOT, "tools", "probe", "lib", name))
    L.votenet_fps_temp_floats.restype = ctypes.c_size_t
    L.votenet_fps_temp_floats.argtypes = [ctypes.c_int, ctypes.c_int]
    L.votenet_farthest_point_sample.argtypes = [ctypes.c_int] * 3 + [ctypes.c_void_p] * 4
    return L


def time_ms(L, x, it=10):
    temp = torch.empty(L.votenet_fps_temp_floats(b, n), dtype=torch.float32, device=dev)
    out = torch.empty((b, m), dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        L.votenet_farthest_point_sample(b, n, m, x.data_ptr(), temp.data_ptr(), out.data_ptr(), st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        L.votenet_farthest_point_sample(b, n, m, x.data_ptr(), temp.data_ptr(), out.data_ptr(), st)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it, temp, out


prod = load(os.path.join("..", "..", "..", "votenet_amd", "lib", "libvotenet_hip.so"))
for kind in ("room", "uniform"):
    x = torch.from_numpy(synth.room_batch(b, n, 1000) if kind == "room" else synth.uniform_batch(b, n, 1000)).to(dev)
    full, _, _ = time_ms(prod, x)
    print("== %s scenes: product kernel %.4f ms per launch (sort 0.065 ms included) = %.3f us = %.0f cycles per round at 2.4 GHz"
          % (kind, full, full * 1e3 / (m - 1), full * 1e-3 / (m - 1) * CLK))
    print("(2) ablations, un-instrumented, ms per launch -> cycles per round removed by each step")
    prev = full
    for a, what in ((1, "touched-bucket updates + bucket arg-max removed"), (2, "... and the box tests + ballot"),
                    (3, "... and the cross-wave exchange + barrier")):
        t, _, _ = time_ms(load("libfps_ablate%d.so" % a), x)
        print("   ablate %d: %.4f ms  (-%4.0f cycles per round)  %s" % (a, t, (prev - t) * 1e-3 / (m - 1) * CLK, what))
        prev = t
    print("   what is left: %.0f cycles per round = loop, wave-winner check, output bookkeeping, launch + prologue (point loads) / %d rounds"
          % (prev * 1e-3 / (m - 1) * CLK, m - 1))
    T = load("libfps_trace.so")
    buf = (ctypes.c_ulonglong * 8)()
    tt, _, _ = time_ms(T, x, it=1)
    T.votenet_fps_trace_read(buf, 1)
    _ = time_ms(T, x, it=1)
    # time_ms ran 3 warm-ups + 1 timed launch since the reset: 4 launches accumulated
    T.votenet_fps_trace_read(buf, 1)
    names = ["loop back-edge + output", "box tests + ballot", "touched buckets", "wave winner", "cross-wave exchange + barrier wait"]
    tot = sum(buf[:5]) / 4.0
    print("(1) s_memtime stamps, instrumented build (%.4f ms per launch): %.0f ticks per round" % (tt, tot / (m - 1)))
    for i, nm in enumerate(names):
        print("   %-36s %7.0f per round  %5.1f%%" % (nm, buf[i] / 4.0 / (m - 1), 100.0 * buf[i] / 4.0 / tot))
