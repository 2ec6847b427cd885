"""How full are the balls?  Mean / distribution of pts_cnt per level on the benchmark's room scenes: slots beyond pts_cnt repeat slot 0
(tf_grouping_g.cu:26-29), i.e. their rows are identical through every layer of the grouped MLP."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [R]
import torch
from votenet_amd import model as VM, synth
dev = torch.device("cuda:0")
net = VM.VoteNetHotPath(dev, seed=0)
x = torch.from_numpy(synth.room_batch(8, 20480, 1000)).to(dev)
tape = []
net.forward(x, tape)
for name, rec in zip(("sa1", "sa2", "sa3", "sa4"), tape[:4]):
    c = rec["pts_cnt"].float()
    k = rec["idx"].shape[2]
    print("%s: K %d  mean pts_cnt %.1f  (%.0f %% of the rows repeat slot 0)  full balls %.0f %%  quartiles %s" % (
        name, k, c.mean().item(), 100 * (1 - c.mean().item() / k), 100 * (c == k).float().mean().item(),
        torch.quantile(c, torch.tensor([0.25, 0.5, 0.75], device=dev)).tolist()))
rec = tape[-1]
c = rec["pts_cnt"].float(); k = rec["idx"].shape[2]
print("proposal: K %d  mean pts_cnt %.1f  (%.0f %% repeat slot 0; untrained votes)" % (k, c.mean().item(), 100 * (1 - c.mean().item() / k)))
# how many grouped rows a layout of g-row pieces keeps (a piece is kept when it holds at least one slot < pts_cnt; the first always)
for name, rec in zip(("sa1", "sa2", "sa3", "sa4"), tape[:4]):
    c = rec["pts_cnt"].float().clamp(min=1)
    k = rec["idx"].shape[2]
    out = []
    for g in (32, 16, 8):
        kept = (torch.ceil(c / g) * g).mean().item()
        out.append("%d-row pieces keep %.0f %%" % (g, 100 * kept / k))
    print("%s: %s; P(cnt<=15) %.2f P(cnt<=31) %.2f P(cnt<=47) %.2f" % (name, ", ".join(out), (c <= 15).float().mean().item(),
          (c <= 31).float().mean().item(), (c <= 47).float().mean().item()))
