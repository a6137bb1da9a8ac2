import torch, time
n = 4 << 30
h = torch.empty(n, dtype=torch.uint8, pin_memory=True); h.fill_(1)
d = torch.empty(n, dtype=torch.uint8, device="cuda")
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter(); d.copy_(h, non_blocking=True); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("H2D pinned one copy GB/s", n / dt / 1e9)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(0, n, 4 << 20):
    d[i:i + (4 << 20)].copy_(h[i:i + (4 << 20)], non_blocking=True)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("H2D pinned 4 MiB pieces GB/s", n / dt / 1e9)
torch.cuda.synchronize(); t0 = time.perf_counter(); h.copy_(d, non_blocking=True); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("D2H GB/s", n / dt / 1e9)
