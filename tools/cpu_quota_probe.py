import os, time, threading, ctypes
for f in ("/sys/fs/cgroup/cpu.max","/sys/fs/cgroup/cpu/cpu.cfs_quota_us","/sys/fs/cgroup/cpu/cpu.cfs_period_us","/sys/fs/cgroup/cpuset.cpus.effective"):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, "n/a")
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
import numpy as np
def work(n, out, k):
    a = np.random.rand(200_000)
    t0=time.perf_counter(); c=0
    while time.perf_counter()-t0 < 1.0:
        a = np.sqrt(a*a+1.0); c+=1
    out[k]=c
for nt in (1,8,16,32,64,128):
    out=[0]*nt
    th=[threading.Thread(target=work,args=(0,out,k)) for k in range(nt)]
    t0=time.perf_counter()
    [t.start() for t in th]; [t.join() for t in th]
    print(nt, "threads: total iterations/s", sum(out)/(time.perf_counter()-t0), "per thread", sum(out)/nt)
