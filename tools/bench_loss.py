import sys, torch
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [R, os.path.join(R, "tests"), os.path.join(R, "tools")]
import loss_ref
from votenet_amd import loss as VL
from bench_mlp_util import timeit
dev = torch.device("cuda:0")
seeds, votes, prop, out, gt = loss_ref.random_case(2, b=8, n=1024, p=256, bb=9)
o = dict(seeds_xyz=torch.from_numpy(seeds).to(dev), votes_xyz=torch.from_numpy(votes).to(dev), proposals_xyz=torch.from_numpy(prop).to(dev), proposals_output=torch.from_numpy(out).to(dev))
g = VL.gt_to_device(gt, dev)
print("loss call: %.3f ms" % timeit(lambda: VL.votenet_loss(o, g), it=50))
