import os, time, json, multiprocessing as mp
def read(p):
    try: return open(p).read().strip()
    except Exception as e: return None
info = {k: read(p) for k, p in {"cpu.max": "/sys/fs/cgroup/cpu.max", "cpu.stat": "/sys/fs/cgroup/cpu.stat", "cpuset.cpus.effective": "/sys/fs/cgroup/cpuset.cpus.effective",
                                "cfs_quota_us": "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "cfs_period_us": "/sys/fs/cgroup/cpu/cpu.cfs_period_us", "memory.max": "/sys/fs/cgroup/memory.max"}.items()}
info["affinity"] = len(os.sched_getaffinity(0))
info["loadavg"] = read("/proc/loadavg")
print(json.dumps(info))
def spin(q, secs):
    t = time.perf_counter(); n = 0
    while time.perf_counter() - t < secs:
        for _ in range(20000): n += 1
    q.put(n)
for n in (1, 2, 4, 8, 16, 32, 64, 128, 256):
    q = mp.Queue(); ps = [mp.Process(target=spin, args=(q, 1.0)) for _ in range(n)]
    t = time.perf_counter()
    for p in ps: p.start()
    tot = sum(q.get() for _ in ps)
    for p in ps: p.join()
    print(n, "procs:", round(tot / 1e6, 1), "M iterations in", round(time.perf_counter() - t, 2), "s", flush=True)
print(read("/sys/fs/cgroup/cpu.stat"))
