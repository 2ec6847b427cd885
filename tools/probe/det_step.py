"""probe: deterministic mode, two replicas from one seed stepping on the same batches -- parameters bit-equal after every step?
(what bench.py --gpus N's self-check asks, without the process group)   python tools/probe/det_step.py [steps]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [R]
import torch
import votenet_amd
from votenet_amd import loss as VL, model as VM, synth, mlp as M
dev = torch.device("cuda:0")
B, n = int(os.environ.get("B", "2")), 20480
votenet_amd.set_deterministic(True)
xs = [torch.from_numpy(synth.room_batch(B, n, s)).to(dev) for s in (1000, 500000)]
gts = [VL.gt_to_device(synth.room_gt(B, n, s), dev) for s in (1000, 500000)]
nets = [VM.VoteNetHotPath(dev, seed=0) for _ in range(2)]
for net in nets:
    net.init_optimizer()
    if os.environ.get("OVERLAP") == "0":  # weight gradients on the chain's own stream
        net.overlap_wgrad = False
# SYNC=a,b,...: a device synchronize at the named places (where does the order between the two streams matter?)
from votenet_amd import pointnet2 as P
SYNC = set(filter(None, os.environ.get("SYNC", "").split(",")))
def _sync_after(obj, name):
    f = getattr(obj, name)
    def g(*a, **k):
        r = f(*a, **k)
        torch.cuda.synchronize()
        return r
    setattr(obj, name, g)
def _sync_before(obj, name):
    f = getattr(obj, name)
    def g(*a, **k):
        torch.cuda.synchronize()
        return f(*a, **k)
    setattr(obj, name, g)
for net in nets:
    if "backward_entry" in SYNC:
        _sync_before(net, "_backward")
    for m in ("proposal", "fp2", "fp1", "sa4", "sa3", "sa2", "sa1"):
        if m in SYNC:
            _sync_after(getattr(net, m), "backward")
        if m + "_before" in SYNC:
            _sync_before(getattr(net, m), "backward")
if "handover" in SYNC:
    _sync_after(P, "_hand_over")
if "step" in SYNC:
    for net in nets:
        _sync_after(net, "train_step")
# TRACE=1: a checksum (the int64 sum of the bit patterns) of every tensor going into and coming out of every votenet_amd.mlp call, per
# replica and step; after a step that differs the first call whose checksums differ is printed
TRACE = os.environ.get("TRACE") == "1"
trace = []
kept = []
keeps = [None, None]
def _sums(objs, out):
    for o in objs:
        if isinstance(o, torch.Tensor) and o.is_cuda and o.numel() > 0:
            t = o.detach()
            t = t if t.is_contiguous() else t.contiguous()
            if t.element_size() == 4:
                out.append(t.view(torch.int32).sum(dtype=torch.int64))
            elif t.element_size() == 8:
                out.append(t.view(torch.int64).sum())
            else:
                out.append(t.view(torch.uint8).sum(dtype=torch.int64))
        elif isinstance(o, (tuple, list)):
            _sums(o, out)
        elif hasattr(o, "scale") and hasattr(o, "shift"):
            _sums([getattr(o, "scale", None), getattr(o, "shift", None)], out)
def _traced(name, f):
    def g(*a, **k):
        ins = []
        _sums(list(a) + list(k.values()), ins)
        r = f(*a, **k)
        outs = []
        _sums([r], outs)
        trace.append((name, ins, outs))
        if name == "pool_dgrad":
            kept.append((r[0].clone(), r[1].clone(), a[10].clone(), a[11].clone(), a, k))  # da, coef of the layer below, arg-max, zsel, the call
        return r
    return g
if TRACE:
    import types
    for nm in dir(M):
        f = getattr(M, nm)
        if isinstance(f, types.FunctionType) and not nm.startswith("_") and f.__module__ == M.__name__ and nm not in ("arena_begin", "arena_end", "half_groups"):
            setattr(M, nm, _traced(nm, f))
bad = 0
traces = [None, None]
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    outs = []
    for r_, net in enumerate(nets):
        del trace[:]
        del kept[:]
        outs.append(net.train_step(xs[i % 2], None, 1, gt=gts[i % 2]))
        traces[r_] = list(trace)
        keeps[r_] = list(kept)
    torch.cuda.synchronize()
    fwd = [k for k in outs[0].keys() if isinstance(outs[0][k], torch.Tensor) and not torch.equal(outs[0][k], outs[1][k])]
    lsame = torch.equal(nets[0].last_losses, nets[1].last_losses)
    same = torch.equal(nets[0].store.flat, nets[1].store.flat)
    gsame = torch.equal(nets[0].store.grad, nets[1].store.grad)
    if not same or not gsame:
        bad += 1
        d = (nets[0].store.grad != nets[1].store.grad)
        names = []
        for name, v in nets[0].store.views.items():
            g0, g1 = nets[0].store.g(name), nets[1].store.g(name)
            if not torch.equal(g0, g1):
                names.append("%s(%d)" % (name, int((g0 != g1).sum())))
        g0, g1 = nets[0].store.grad, nets[1].store.grad
        rel = float((g0 - g1).abs().max() / g0.abs().max())
        if TRACE:
            t0, t1 = traces
            print("   %d / %d traced calls" % (len(t0), len(t1)))
            shown = 0
            for ci, (c0, c1) in enumerate(zip(t0, t1)):
                i0, i1 = [int(v) for v in c0[1]], [int(v) for v in c1[1]]
                o0, o1 = [int(v) for v in c0[2]], [int(v) for v in c1[2]]
                if i0 != i1 or o0 != o1:
                    print("   call %d %s: inputs differ at %s, outputs differ at %s" % (ci, c0[0], [j for j in range(len(i0)) if i0[j] != i1[j]], [j for j in range(len(o0)) if o0[j] != o1[j]]))
                    shown += 1
                    if shown >= 6:
                        break
        if TRACE and keeps[0]:
            (d0, c0, am, zs, a_, k_), (d1, c1, _, _, _, _) = keeps[0][0], keeps[1][0]
            ne = (d0 != d1)
            rows = ne.any(dim=1).nonzero().flatten()
            print("   pool_dgrad: %d of %d rows differ (%d values), max |diff| %.3g of max %.3g; coef equal %s" % (rows.numel(), d0.shape[0], int(ne.sum()), float((d0 - d1).abs().max()), float(d0.abs().max()), torch.equal(c0, c1)))
            print("   rows: %s" % rows[:24].tolist())
            print("   slots (row %% 64): %s" % sorted(set((rows % 64).tolist()))[:64])
            grp = (rows // 64).unique()
            print("   groups: %d distinct, first %s" % (grp.numel(), grp[:16].tolist()))
            if rows.numel():
                r0_ = int(rows[0]); g0_ = r0_ // 64
                print("   row %d: replica 0 %s" % (r0_, d0[r0_, :8].tolist()))
                print("   row %d: replica 1 %s" % (r0_, d1[r0_, :8].tolist()))
                for rr_ in rows[:6].tolist():
                    cols = ne[rr_].nonzero().flatten().tolist()
                    print("   row %d (group %d slot %d): columns %s" % (rr_, rr_ // 64, rr_ % 64, cols))
                    print("      replica 0 %s" % ["%.6g" % v for v in d0[rr_, cols[:6]].tolist()])
                    print("      replica 1 %s" % ["%.6g" % v for v in d1[rr_, cols[:6]].tolist()])
                    print("      arg-max hits of the group's slots 0..5: %s" % [(int((am[rr_ // 64] == t_).sum())) for t_ in range(6)])
                # the truth for the first rows that differ, in float64: dense part + the listed channels' rows of W^T
                xz_, isc, ish, irelu, w_, b_, wT_, coef_, relu_, gout_ = a_[:10]
                mm_ = k_.get("mm")
                dense = M.linear_dense(xz_, mm_[:xz_.shape[1]], mm_[xz_.shape[1]], isc, ish, irelu, want_stats=False)[0]
                co = w_.shape[1]
                cA, cS, cH = coef_[:co].double(), coef_[3 * co:4 * co].double(), coef_[4 * co:5 * co].double()
                for rr_ in rows[:4].tolist():
                    g_, t_ = rr_ // 64, rr_ % 64
                    gg = gout_[g_].double().clone()
                    if relu_:
                        gg[~((zs[g_].double() * cS + cH) > 0)] = 0
                    v_ = cA * gg
                    sel = (am[g_] == t_) & (v_ != 0)
                    truth = dense[rr_].double() + (v_[sel].unsqueeze(1) * wT_[sel].double()).sum(0)
                    cols = ne[rr_].nonzero().flatten()
                    print("   row %d: %d listed channels; at the differing columns  replica0 - truth %s" % (rr_, int(sel.sum()), ["%.3g" % v for v in (d0[rr_, cols].double() - truth[cols])[:5].tolist()]))
                    print("        replica1 - truth %s   (other columns: max |replica - truth| %.3g)" % (["%.3g" % v for v in (d1[rr_, cols].double() - truth[cols])[:5].tolist()], float((d0[rr_].double() - truth).abs()[~ne[rr_]].max())))
                    bad_r = d0 if (d0[rr_, cols].double() - truth[cols]).abs().max() > (d1[rr_, cols].double() - truth[cols]).abs().max() else d1
                    err = bad_r[rr_, cols].double() - truth[cols]
                    # is the error one listed channel's contribution (missing or twice)?
                    ch = sel.nonzero().flatten()
                    contrib = v_[ch].unsqueeze(1) * wT_[ch][:, cols].double()  # (n, 16)
                    dist = ((contrib - err.unsqueeze(0)).abs().max(1)[0], (contrib + err.unsqueeze(0)).abs().max(1)[0])
                    print("        error vs one channel's contribution: min |err - c| %.3g, min |err + c| %.3g (|err| max %.3g); err vs dense part %.3g" % (float(dist[0].min()) if ch.numel() else -1, float(dist[1].min()) if ch.numel() else -1, float(err.abs().max()), float((bad_r[rr_, cols].double() - dense[rr_, cols].double()).abs().max())))
                print("   group %d: channels whose arg-max is this slot: %d" % (g0_, int((am[g0_] == r0_ % 64).sum())))
        print("   forward outputs that differ: %s; losses equal %s" % (fwd, lsame))
        print("step %d: params equal %s, grads equal %s; %d grad values differ, max |diff| / max |g| = %.3g: %s" % (i, same, gsame, int(d.sum()), rel, " ".join(names[:6])))
        print("   %d of %d tensors differ; equal ones: %s" % (len(names), len(nets[0].store.views), " ".join(nm for nm in nets[0].store.views if torch.equal(nets[0].store.g(nm), nets[1].store.g(nm)))))
        nets[1].store.flat.copy_(nets[0].store.flat); nets[1].store.params_changed()
        nets[1]._m.copy_(nets[0]._m); nets[1]._v.copy_(nets[0]._v)
print("%d steps with a difference" % bad)
