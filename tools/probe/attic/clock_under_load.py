"""probe: the shader clock rocm-smi reports while (a) the pipelined train step loops, (b) only the sa1 sampling launch loops.
A child process polls `rocm-smi --showclocks`; this process keeps the GPU busy for ~12 s per mode."""
import os, subprocess, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [R]
import torch
from votenet_amd import loss as VL, model as VM, synth, tf_sampling

def poll(tag, seconds):
    code = ("import subprocess,time,re\n"
            "t0=time.time(); vals=[]\n"
            "while time.time()-t0 < %f:\n"
            "    o=subprocess.run(['/opt/rocm/bin/rocm-smi','--showclocks'],capture_output=True,text=True).stdout\n"
            "    m=re.search(r'sclk clock level: \\d+: \\((\\d+)Mhz\\)', o)\n"
            "    vals.append(int(m.group(1)) if m else -1)\n"
            "    time.sleep(0.3)\n"
            "print('%s sclk MHz:', vals)\n") % (seconds, tag)
    return subprocess.Popen([sys.executable, "-c", code])

dev = torch.device("cuda:0")
xs = [torch.from_numpy(synth.room_batch(8, 20480, s)).to(dev) for s in (1000, 500000, 900000)]
gts = [VL.gt_to_device(synth.room_gt(8, 20480, s), dev) for s in (1000, 500000, 900000)]
net = VM.VoteNetHotPath(dev, seed=0)
for i in range(10):
    net.train_step(xs[i % 3], gt=gts[i % 3], next_x=[xs[(i + 1) % 3]])
torch.cuda.synchronize()
p = poll("train step looping:", 10.0)
t0 = time.time(); i = 0
while time.time() - t0 < 12.0:
    net.train_step(xs[i % 3], gt=gts[i % 3], next_x=[xs[(i + 1) % 3]]); i += 1
    if i % 50 == 0: torch.cuda.synchronize()
torch.cuda.synchronize(); p.wait()
net.drop_graphs()
p = poll("sa1 sampling alone:", 10.0)
t0 = time.time()
while time.time() - t0 < 12.0:
    for _ in range(20): tf_sampling.farthest_point_sample(2048, xs[0])
    torch.cuda.synchronize()
p.wait()
