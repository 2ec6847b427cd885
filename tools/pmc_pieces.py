"""The forward + backward of ONE set-abstraction level on the piece layout (csrc/half.hip), REP times on one stream, for counter
collection (GPU box only): real geometry and activations of room scenes (8 x 20480), sa1 (narrow first layer, 16384 balls) or sa2
(assembled first layer, 8192 balls).    python tools/pmc_pieces.py sa2
tools/pmc_pieces.sh wraps the rocprofv3 passes; tools/pmc_pieces_summary.py composes profiles/rNN_pmc_pieces.txt."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [R]
import torch
from votenet_amd import model as VM, synth, pointnet2 as P
level = sys.argv[1] if len(sys.argv) > 1 else "sa2"
REP = int(os.environ.get("REP", "4"))
dev = torch.device("cuda:0")
net = VM.VoteNetHotPath(dev, seed=0)
x = torch.from_numpy(synth.room_batch(8, 20480, 7)).to(dev)
tape = []
net.forward(x, tape)
rec = tape[{"sa1": 0, "sa2": 1, "sa3": 2, "sa4": 3}[level]]
mod = rec["module"]
xyz, pts = rec["xyz"], rec["points"]
geom = mod.geometry(xyz, points=pts)
g = torch.Generator().manual_seed(1)
gout = torch.randn(rec["b"], mod.npoint, mod.mlp[-1].cout, generator=g).to(dev)
net.store.grad.zero_()
torch.cuda.synchronize()
half = geom[-1].resolve()
print("%s: %d balls, %d pieces = %d compact rows (%.0f %% of %d)" % (level, half.G, half.nh, half.rows, 100.0 * half.rows / (half.G * 64), half.G * 64))
for _ in range(REP):
    t = []
    mod.forward(xyz, pts, tape=t, geom=geom)
    mod.backward(t[0], gout, need_feat_grad=pts is not xyz)
    P.wgrad_join()
torch.cuda.synchronize()
