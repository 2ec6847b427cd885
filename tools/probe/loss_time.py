"""probe: the loss launch (count kernel + loss kernel) alone on the GPU at the headline shapes, library VARIANT as variant_step.py."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [R]
import torch
from votenet_amd import _lib as L_
if os.environ.get("VARIANT"):
    L_._LIB_PATH = os.path.join(R, "tools", "probe", "lib", "libvotenet_%s.so" % os.environ["VARIANT"])
from votenet_amd import loss as VL, synth
dev = torch.device("cuda:0")
torch.manual_seed(0)
gt = VL.gt_to_device(synth.room_gt(8, 20480, 1000), dev)
out = dict(seeds_xyz=torch.randn(8, 1024, 3, device=dev) + 1, votes_xyz=torch.randn(8, 1024, 3, device=dev) + 1,
           proposals_xyz=gt["bboxes_xyz"].repeat(1, 26, 1)[:, :256].contiguous() + 0.2 * torch.randn(8, 256, 3, device=dev),
           proposals_output=torch.randn(8, 256, 79, device=dev))
for _ in range(5): VL.votenet_loss(out, gt)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): VL.votenet_loss(out, gt)
e1.record(); torch.cuda.synchronize()
print("loss launch (fill + count + loss kernels) %-8s %.1f us" % (os.environ.get("VARIANT") or "(built)", e0.elapsed_time(e1) / 50 * 1e3))
