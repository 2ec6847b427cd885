"""Where the host thread of a rank runs.  A train step is ~160 launches of 10-20 us each: the Python thread, the HIP runtime's helper
threads and the driver share cache lines all the time, and the step is as fast as the host can enqueue it on its small-kernel stretch
(DESIGN_HISTORY.md 5).  Left to the scheduler on a 2 x 64-core host the enqueue of a step takes 3.1 ms; confined to eight cores of the GPU's NUMA
node 2.75 ms (four cores: 3.15, one: 3.5 -- the helper threads need room; the other socket: 3.1; tools/cpu_issue_time.py under taskset).

No torch, no HIP: this runs before anything touches the GPU (threads created later inherit the mask).  `from votenet_amd import hostpin`
would run the package's __init__ and import torch first -- threads torch starts at import keep the wide mask -- so bench.py and
tools/train_eval.py load THIS FILE by path (`load_standalone`'s recipe) before they import anything else."""
import os


def _cpulist(text):
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out.extend(range(int(a), int(b or a) + 1))
    return out


def gpu_numa_nodes(root="/sys/class/kfd/kfd/topology/nodes", pci="/sys/bus/pci/devices"):
    """NUMA node of every GPU in KFD topology order (the order HIP enumerates them in), -1 where unknown."""
    nodes = []
    try:
        names = sorted(os.listdir(root), key=lambda v: int(v) if v.isdigit() else 0)
    except OSError:
        return nodes
    for nd in names:
        try:
            props = dict(ln.split(None, 1) for ln in open(os.path.join(root, nd, "properties")).read().splitlines() if " " in ln)
        except OSError:
            continue
        if int(props.get("simd_count", "0").strip() or 0) <= 0:
            continue  # a CPU node
        numa = -1
        try:
            loc, dom = int(props.get("location_id", "0")), int(props.get("domain", "0"))
            bdf = "%04x:%02x:%02x.%x" % (dom, (loc >> 8) & 0xFF, (loc >> 3) & 0x1F, loc & 7)
            numa = int(open(os.path.join(pci, bdf, "numa_node")).read().strip())
        except (OSError, ValueError):
            pass
        nodes.append(numa)
    return nodes


def physical_gpu(local_rank, ngpus, environ=None):
    """Device `local_rank` of this process -> its index in KFD topology order.  ROCR_VISIBLE_DEVICES filters (and reorders) what the
    runtime sees; HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES then index into THAT list.  An entry that is not a plain index (a GPU-uuid)
    or is out of range -> None with a warning: the rank is then not pinned at all rather than to some other GPU's NUMA node."""
    import warnings
    env = os.environ if environ is None else environ
    devs = list(range(ngpus))
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = env.get(var)
        if not v:
            continue
        if var == "CUDA_VISIBLE_DEVICES" and env.get("HIP_VISIBLE_DEVICES"):
            continue  # HIP reads one of the two; HIP_VISIBLE_DEVICES wins
        picked = []
        for t in (t.strip() for t in v.split(",")):
            if not t.isdigit() or int(t) >= len(devs):
                if t and not t.isdigit():
                    warnings.warn("hostpin: %s=%s names a device by something other than an index: host threads left unpinned" % (var, v))
                    return None
                break  # the runtimes stop at the first invalid index
            picked.append(devs[int(t)])
        devs = picked
    if not 0 <= local_rank < len(devs):
        return None
    return devs[local_rank]


def pin(local_rank=0, cores=8, env="VOTENET_NO_PIN"):
    """Confine this process to `cores` consecutive CPUs of the NUMA node of GPU `local_rank` (rank r takes the r-th block of its node,
    so ranks sharing a node do not share cores).  A no-op when the mask is already narrow (a launcher or a container chose), the
    environment variable `env` is set, or the topology cannot be read.  -> the CPUs chosen, or None."""
    if os.environ.get(env) or not hasattr(os, "sched_setaffinity"):
        return None
    try:
        have = sorted(os.sched_getaffinity(0))
        if len(have) < 2 * cores:
            return None
        numas = gpu_numa_nodes()
        phys = physical_gpu(local_rank, len(numas))
        if phys is None:
            return None
        node = numas[phys] if 0 <= phys < len(numas) else -1
        cand = have
        if node >= 0:
            on_node = set(_cpulist(open("/sys/devices/system/node/node%d/cpulist" % node).read()))
            cand = [c for c in have if c in on_node] or have
        same = sum(1 for n in numas[:phys] if n == node)  # GPUs before this one on the same node: their ranks take the blocks before
        blocks = max(1, len(cand) // cores)
        k = (1 + same) % blocks  # block 0 holds CPU 0, where the kernel's own housekeeping tends to land: start at block 1
        chosen = cand[k * cores:(k + 1) * cores]
        if len(chosen) < cores:
            return None
        os.sched_setaffinity(0, chosen)
        return chosen
    except (OSError, ValueError, IndexError):
        return None


def unpin(mask):
    """Give EVERY thread of this process the CPU set `mask` again (what os.sched_getaffinity(0) returned before pin()): threads created
    while the process was confined -- an OpenMP pool, the runtime's helpers -- keep the narrow mask otherwise."""
    if not mask or not hasattr(os, "sched_setaffinity"):
        return
    try:
        tids = [int(t) for t in os.listdir("/proc/self/task")]
    except OSError:
        tids = [0]
    for t in tids:
        try:
            os.sched_setaffinity(t, mask)
        except OSError:
            pass
