import time, torch, numpy as np
torch.cuda.init()
d = torch.empty(100_000_000, dtype=torch.uint8, device="cuda")
for n in (22_000_000, 100_000_000):
    t0=time.perf_counter(); p = torch.empty(n, dtype=torch.uint8, pin_memory=True); t1=time.perf_counter()
    print("pinned alloc", n, round((t1-t0)*1e3,1), "ms")
    a = np.zeros(n, dtype=np.uint8); a[:] = 1
    t0=time.perf_counter(); d[:n].copy_(torch.from_numpy(a)); torch.cuda.synchronize(); t1=time.perf_counter()
    print("H2D pageable", n, round((t1-t0)*1e3,1), "ms", round(n/1e9/(t1-t0),1), "GB/s")
    p[:] = 1
    t0=time.perf_counter(); d[:n].copy_(p, non_blocking=True); torch.cuda.synchronize(); t1=time.perf_counter()
    print("H2D pinned", n, round((t1-t0)*1e3,1), "ms", round(n/1e9/(t1-t0),1), "GB/s")
    t0=time.perf_counter(); p.copy_(d[:n], non_blocking=True); torch.cuda.synchronize(); t1=time.perf_counter()
    print("D2H pinned", n, round((t1-t0)*1e3,1), "ms", round(n/1e9/(t1-t0),1), "GB/s")
    b = torch.from_numpy(a)
    t0=time.perf_counter(); b.copy_(d[:n]); torch.cuda.synchronize(); t1=time.perf_counter()
    print("D2H pageable", n, round((t1-t0)*1e3,1), "ms", round(n/1e9/(t1-t0),1), "GB/s")
    t0=time.perf_counter(); c = a.copy(); t1=time.perf_counter()
    print("host memcpy", n, round((t1-t0)*1e3,1), "ms")
    del p
