"""FPS alone on the GPU: sa1 (8 x 20480 -> 2048, room and uniform scenes), sa2's 2048 -> 1024 and the dense scan's 4 x 80000 -> 2048.  ms per call."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from votenet_amd import _lib as L_
if os.environ.get("VARIANT"):
    L_._LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libvotenet_%s.so" % os.environ["VARIANT"])
from votenet_amd import synth, tf_sampling
dev = torch.device("cuda:0")
def timeit(fn, it=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
room = torch.from_numpy(synth.room_batch(8, 20480, 1000)).to(dev)
uni = torch.from_numpy(synth.uniform_batch(8, 20480, 1000)).to(dev)
big = torch.from_numpy(synth.room_batch(4, 80000, 7)).to(dev)
small = room[:, :2048].contiguous()
print("sa1 room %.4f  uniform %.4f  2048->1024 %.4f  80000->2048 %.4f" % (
    timeit(lambda: tf_sampling.farthest_point_sample(2048, room)), timeit(lambda: tf_sampling.farthest_point_sample(2048, uni)),
    timeit(lambda: tf_sampling.farthest_point_sample(1024, small)), timeit(lambda: tf_sampling.farthest_point_sample(2048, big), it=4, warm=1)))
