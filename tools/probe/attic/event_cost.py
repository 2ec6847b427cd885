"""What does a hand-over to a side stream cost the MAIN queue?  A chain of small dependent kernels, with and without an event record
(+ side-stream wait and a side kernel) between them (scratch, GPU box)."""
import torch, time
dev = torch.device("cuda:0")
N = int(__import__("os").environ.get("N", "4096")); x = torch.zeros(N, device=dev); y = torch.zeros(4096, device=dev)
side = torch.cuda.Stream(device=dev)
def chain(n, mode):
    for i in range(n):
        x.add_(1.0)
        if mode >= 1:
            ev = torch.cuda.Event(); ev.record()
            if mode >= 2:
                side.wait_event(ev)
                if mode >= 3:
                    with torch.cuda.stream(side):
                        y.add_(1.0)
def t(mode, n=400):
    chain(50, mode); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record(); chain(n, mode); e1.record(); h = time.perf_counter() - t0; torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3, h / n * 1e6
for mode, name in ((0, "kernels only"), (1, "+ event record"), (2, "+ side.wait_event"), (3, "+ side kernel")):
    g, h = t(mode)
    print("%-20s GPU %.2f us per link, host %.2f us per link" % (name, g, h))
