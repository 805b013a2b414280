import torch, time
n = 150_000_000          # 1.2 GB of doubles
a = torch.empty(n, dtype=torch.float64, device='cuda'); b = torch.ones(n, dtype=torch.float64, device='cuda')
def t(f, k=20):
    f(); torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(k): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/k
dt = t(lambda: a.copy_(b)); print(f"copy 1.2 GB: {dt*1e3:.3f} ms -> {2*n*8/dt/1e12:.2f} TB/s (read+write)")
dt = t(lambda: torch.add(a, b, out=a)); print(f"a += b    : {dt*1e3:.3f} ms -> {3*n*8/dt/1e12:.2f} TB/s (2 reads + 1 write)")
c = torch.empty(n//2, dtype=torch.float64, device='cuda')
dt = t(lambda: torch.sum(b)); print(f"sum(b)    : {dt*1e3:.3f} ms -> {n*8/dt/1e12:.2f} TB/s (read only)")
dt = t(lambda: a.fill_(1.0)); print(f"fill      : {dt*1e3:.3f} ms -> {n*8/dt/1e12:.2f} TB/s (write only)")
