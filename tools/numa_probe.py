"""Where does pinned host memory land, and what does that do to the link rate?  (The GPU boxes have two sockets; the GPU hangs off
one of them.)  For every placement policy: allocate 82 MB with spx_host_alloc (hipHostMalloc), read the pages' nodes from
/proc/self/numa_maps, time 30 host-to-device and 30 device-to-host copies.  python tools/numa_probe.py"""
import ctypes as C
import os
import re
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from speedy_amd._lib import lib  # noqa: E402

L = lib()
libc = C.CDLL(None, use_errno=True)
SYS_set_mempolicy = 238
MPOL_DEFAULT, MPOL_PREFERRED, MPOL_BIND = 0, 1, 2
NBYTES = 256 * 160000 * 2


def set_policy(mode, node=None):
    if node is None:
        return libc.syscall(SYS_set_mempolicy, MPOL_DEFAULT, None, 0)
    mask = C.c_ulong(1 << node)
    return libc.syscall(SYS_set_mempolicy, mode, C.byref(mask), 64)


def nodes_of(addr):
    for line in open("/proc/self/numa_maps"):
        if line.startswith("%x " % addr) or line.startswith("%012x " % addr):
            return " ".join(re.findall(r"N\d+=\d+", line)) or line.strip()[:120]
    return "?"


def gpu_node():
    try:
        p = torch.cuda.get_device_properties(0)
        bus = "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
        return int(open("/sys/bus/pci/devices/%s/numa_node" % bus).read()), bus
    except Exception as e:  # noqa: BLE001
        return -1, str(e)


def node_cpus(node):
    s = open("/sys/devices/system/node/node%d/cpulist" % node).read().strip()
    cpus = []
    for part in s.split(","):
        a, _, b = part.partition("-")
        cpus += list(range(int(a), int(b or a) + 1))
    return cpus


torch.cuda.init()
d = torch.empty(NBYTES // 2, dtype=torch.int16, device="cuda")
gn, bus = gpu_node()
print("GPU numa node %d (%s); this thread on cpu %d" % (gn, bus, libc.sched_getcpu()))
all_cpus = sorted(os.sched_getaffinity(0))
cases = [("default", None, None)]
for node in (0, 1):
    if os.path.exists("/sys/devices/system/node/node%d" % node):
        cases += [("prefer node %d" % node, node, None), ("cpus of node %d" % node, None, node)]
for name, pol_node, cpu_node in cases * 2:
    os.sched_setaffinity(0, node_cpus(cpu_node) if cpu_node is not None else all_cpus)
    set_policy(MPOL_PREFERRED, pol_node)
    p = L.spx_host_alloc(NBYTES)
    C.memset(p, 1, NBYTES)
    where = nodes_of(p)
    st = torch.cuda.current_stream().cuda_stream
    res = []
    for fn, a, b in ((L.spx_copy_to_device, d.data_ptr(), p), (L.spx_copy_to_host, p, d.data_ptr())):
        for _ in range(5):
            fn(a, b, NBYTES, st)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30):
            fn(a, b, NBYTES, st)
        torch.cuda.synchronize()
        res.append(NBYTES / ((time.perf_counter() - t0) / 30) / 1e9)
    print("%-16s cpu %3d  pages %-24s  H2D %.1f GB/s  D2H %.1f GB/s" % (name, libc.sched_getcpu(), where, res[0], res[1]), flush=True)
    L.spx_host_free(p)
    set_policy(MPOL_DEFAULT)
os.sched_setaffinity(0, all_cpus)
