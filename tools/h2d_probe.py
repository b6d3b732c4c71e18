import torch, time
x = torch.empty(1<<30, dtype=torch.uint8).pin_memory()
d = torch.empty(1<<30, dtype=torch.uint8, device="cuda")
for n_streams in (1, 2, 4):
    ss = [torch.cuda.Stream() for _ in range(n_streams)]
    torch.cuda.synchronize()
    best = 0
    for rep in range(3):
        t0 = time.perf_counter()
        piece = (1<<30) // n_streams
        for i, s in enumerate(ss):
            with torch.cuda.stream(s):
                d[i*piece:(i+1)*piece].copy_(x[i*piece:(i+1)*piece], non_blocking=True)
        torch.cuda.synchronize()
        best = max(best, (1<<30) / (time.perf_counter() - t0) / 1e9)
    print("streams", n_streams, "GB/s %.1f" % best)
# small pieces of 8 MiB on 4 streams
ss = [torch.cuda.Stream() for _ in range(4)]
P = 8 << 20
torch.cuda.synchronize(); t0 = time.perf_counter()
for k in range((1<<30)//P):
    with torch.cuda.stream(ss[k % 4]):
        d[k*P:(k+1)*P].copy_(x[k*P:(k+1)*P], non_blocking=True)
torch.cuda.synchronize(); print("8MiB pieces x4 streams GB/s %.1f" % ((1<<30)/(time.perf_counter()-t0)/1e9))
