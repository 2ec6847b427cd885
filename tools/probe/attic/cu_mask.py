"""CU-masked HIP streams: the geometry chains (FPS: one 768-thread workgroup per scene for 1.7 ms) on a few CUs of their own, the
GEMM chain on the rest, so that no persistent GEMM workgroup shares a CU with an FPS workgroup (scratch probe, GPU box)."""
import ctypes, os, sys, time, gc
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [R]
import torch
from votenet_amd import model as VM, synth
hip = ctypes.CDLL("libamdhip64.so")
def masked_stream(bits):
    words = (ctypes.c_uint32 * 8)(*[(bits >> (32 * i)) & 0xffffffff for i in range(8)])
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value)
dev = torch.device("cuda:0")
xs = [torch.from_numpy(synth.room_batch(8, 20480, s)).to(dev) for s in (1000, 500000, 900000)]
ALL = (1 << 256) - 1
def trial(name, geo_bits, main_bits):
    net = VM.VoteNetHotPath(dev, seed=0)
    if geo_bits is not None:
        net._side = masked_stream(geo_bits[0])
        net._pf_streams = [net._side, masked_stream(geo_bits[1])]
        net._pf_turn = 0
    ms = masked_stream(main_bits) if main_bits is not None else torch.cuda.current_stream()
    def run(k):
        with torch.cuda.stream(ms):
            for i in range(k):
                net.forward(xs[i % 3], next_x=[xs[(i + 1) % 3], xs[(i + 2) % 3]])
    run(9); torch.cuda.synchronize(); gc.disable()
    t0 = time.perf_counter(); run(60); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 60 * 1e3; gc.enable()
    print("%-58s %.3f ms per forward" % (name, dt), flush=True)
trial("no masks", None, None)
def every(step, off, count):  # `count` CUs: off, off+step, ...
    b = 0
    for i in range(count):
        b |= 1 << (off + i * step)
    return b
for step, label in ((1, "CUs 0..7 / 8..15"), (32, "one CU per XCD-sized stride: 0,32,.. / 1,33,.."), (8, "stride 8")):
    g0, g1 = every(step, 0, 8), every(step, 1 if step > 1 else 8, 8)
    trial("geometry on 8+8 CUs (%s), GEMMs on the rest" % label, (g0, g1), ALL & ~(g0 | g1))
    trial("geometry on 8+8 CUs (%s), GEMMs unmasked" % label, (g0, g1), None)
