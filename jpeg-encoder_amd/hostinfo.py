"""Where the host side of a rank lives: NUMA nodes, their CPUs, the node of a GPU's PCIe root complex, and the node the
pages of a buffer were placed on (Linux `move_pages` in query mode).  bench.py and the diagnostic tools report these
beside every host-fed figure so that a slow run can be attributed (remote source pages, workers on the far socket)
from its own record.  Pure Python + ctypes; every query degrades to None where the kernel or sysfs does not answer."""
import ctypes
import os
import re

_SYS_move_pages = 279            # x86_64
_PAGE = os.sysconf("SC_PAGE_SIZE") if hasattr(os, "sysconf") else 4096


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def parse_cpulist(text):
    cpus = set()
    for part in (text or "").split(","):
        part = part.strip()
        if not part:
            continue
        a, _, b = part.partition("-")
        try:
            lo, hi = int(a), int(b or a)
        except ValueError:
            continue
        cpus.update(range(lo, hi + 1))
    return cpus


def numa_nodes():
    """{node: sorted list of cpus} from sysfs ({} where there is no NUMA information)."""
    out = {}
    base = "/sys/devices/system/node"
    try:
        names = os.listdir(base)
    except OSError:
        return out
    for name in names:
        m = re.fullmatch(r"node(\d+)", name)
        if m:
            out[int(m.group(1))] = sorted(parse_cpulist(_read(f"{base}/{name}/cpulist")))
    return out


def node_of_cpu(cpu, nodes=None):
    for n, cpus in (nodes or numa_nodes()).items():
        if cpu in cpus:
            return n
    return None


def gpu_numa_node(pci_bus_id):
    """NUMA node of the PCI device 'dddd:bb:dd.f' (None when sysfs reports -1 or nothing)."""
    if not pci_bus_id:
        return None
    txt = _read(f"/sys/bus/pci/devices/{pci_bus_id.lower()}/numa_node")
    try:
        n = int(txt)
    except (TypeError, ValueError):
        return None
    return n if n >= 0 else None


def torch_gpu_bus_id(torch, index):
    try:
        p = torch.cuda.get_device_properties(index)
        return f"{getattr(p, 'pci_domain_id', 0):04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
    except Exception:
        return None


def pages_nodes(address, nbytes, samples=64):
    """{node: count} over `samples` pages spread evenly over [address, address + nbytes): where the kernel placed them
    (-errno entries, e.g. pages never touched, are counted under the key 'unplaced')."""
    if not address or nbytes <= 0:
        return None
    first = address // _PAGE
    last = (address + nbytes - 1) // _PAGE
    npages = last - first + 1
    n = min(samples, npages)
    pages = (ctypes.c_void_p * n)(*[(first + (i * npages) // n) * _PAGE for i in range(n)])
    status = (ctypes.c_int * n)()
    try:
        libc = ctypes.CDLL(None, use_errno=True)
        rc = libc.syscall(_SYS_move_pages, 0, ctypes.c_ulong(n), pages, None, status, 0)
    except Exception:
        return None
    if rc != 0:
        return None
    out = {}
    for s in status:
        key = int(s) if s >= 0 else "unplaced"
        out[key] = out.get(key, 0) + 1
    return out


def array_nodes(arr, samples=64):
    """pages_nodes of a numpy array's buffer."""
    return pages_nodes(arr.ctypes.data, arr.nbytes, samples)


def merge_counts(list_of_counts):
    out = {}
    for c in list_of_counts:
        for k, v in (c or {}).items():
            out[k] = out.get(k, 0) + v
    return {str(k): v for k, v in sorted(out.items(), key=lambda kv: str(kv[0]))}


def affinity_summary(nodes=None):
    """How many of the CPUs this thread may run on lie on each node: {'cpus': N, 'per_node': {node: count}}."""
    nodes = nodes if nodes is not None else numa_nodes()
    try:
        allowed = os.sched_getaffinity(0)
    except (AttributeError, OSError):
        return None
    per = {}
    for n, cpus in nodes.items():
        k = len(allowed.intersection(cpus))
        if k:
            per[str(n)] = k
    return {"cpus": len(allowed), "per_node": per}


def host_summary(torch=None, gpu_index=0):
    nodes = numa_nodes()
    model = None
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    mem = {}
    for n in nodes:
        txt = _read(f"/sys/devices/system/node/node{n}/meminfo") or ""
        m = re.search(r"MemTotal:\s+(\d+) kB", txt)
        f = re.search(r"MemFree:\s+(\d+) kB", txt)
        if m:
            mem[str(n)] = {"total_GB": round(int(m.group(1)) / 1e6, 1), "free_GB": round(int(f.group(1)) / 1e6, 1) if f else None}
    bus = torch_gpu_bus_id(torch, gpu_index) if torch is not None else None
    return {"cpu_model": model, "logical_cpus": os.cpu_count(), "numa_nodes": {str(n): f"{len(c)} cpus" for n, c in sorted(nodes.items())},
            "node_memory": mem, "gpu_pci_bus_id": bus, "gpu_numa_node": gpu_numa_node(bus), "affinity": affinity_summary(nodes),
            "this_thread_cpu_node": node_of_cpu(_current_cpu(), nodes),
            "transparent_hugepage": _read("/sys/kernel/mm/transparent_hugepage/enabled"),
            "numa_balancing": _read("/proc/sys/kernel/numa_balancing")}


def _current_cpu():
    try:
        libc = ctypes.CDLL(None)
        return int(libc.sched_getcpu())
    except Exception:
        return None


# ---- CPU budget of a rank (the host side is the limiter of the 8-GPU batch) ------------------------------------------
def cpu_quota():
    """CPUs' worth of run time the container's cgroup allows (cpu.max / cfs quota): None = no limit.  The library reads the same
    files (csrc/host_internal.h usable_cpus)."""
    try:                                                           # cgroup v2
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max" and float(period) > 0:
            return float(quota) / float(period)
        return None
    except (OSError, ValueError):
        pass
    try:                                                           # cgroup v1
        quota = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        period = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return quota / period if quota > 0 and period > 0 else None
    except (OSError, ValueError):
        return None


def usable_cpus():
    """CPUs this process may keep busy at once: the affinity mask, cut by the cgroup quota."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    q = cpu_quota()
    if q:
        n = min(n, max(1, int(q + 0.999)))
    return max(n, 1)


def rank_cpu_share(local_world_size=None):
    """Host threads ONE rank may keep busy when `local_world_size` processes (default: LOCAL_WORLD_SIZE, which torchrun exports
    before any GPU call; 1 without it) share this host's affinity mask and CPU quota: the equal share, at least 1.  What a rank
    passes to Encoder.set_batch_workers - the library sizes its pools by the whole quota, it cannot see its neighbours."""
    if local_world_size is None:
        try:
            local_world_size = int(os.environ.get("LOCAL_WORLD_SIZE", "1"))
        except ValueError:
            local_world_size = 1
    return max(1, usable_cpus() // max(int(local_world_size), 1))


def cpu_stat():
    """(CPU seconds used, CFS periods throttled) so far: of the cgroup where cpu.stat is readable (the whole container), else of
    this process (throttled = None)."""
    try:
        d = {}
        for line in open("/sys/fs/cgroup/cpu.stat"):
            k, _, v = line.partition(" ")
            d[k] = int(v)
        return d["usage_usec"] / 1e6, d.get("nr_throttled", 0)
    except (OSError, ValueError, KeyError):
        t = os.times()
        return t.user + t.system, None


def thread_cpu_times():
    """{tid: (comm, user seconds, system seconds)} of this process's threads (/proc/self/task): which threads a busy CPU belongs to."""
    out = {}
    tick = os.sysconf("SC_CLK_TCK")
    try:
        tids = os.listdir("/proc/self/task")
    except OSError:
        return out
    for tid in tids:
        try:
            txt = open(f"/proc/self/task/{tid}/stat").read()
        except OSError:
            continue
        r = txt.rfind(")")
        comm = txt[txt.find("(") + 1:r]
        f = txt[r + 2:].split()
        out[int(tid)] = (comm, int(f[11]) / tick, int(f[12]) / tick)
    return out
