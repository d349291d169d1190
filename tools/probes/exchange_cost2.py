"""Which call of the exchange pattern blocks the host?  Pure torch streams/events (GPU box)."""
import time, torch
dev = torch.device("cuda:0")
sA, cS = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
x = torch.zeros(1 << 20, device=dev)
counts = torch.zeros(256, dtype=torch.int32, device=dev)
out = torch.zeros(256, dtype=torch.int32, device=dev)
ready = torch.cuda.Event()
done = [torch.cuda.Event() for _ in range(4)]
N = 300
def run(name, use_copy, use_kernel_on_c, heavy):
    torch.cuda.synchronize()
    acc = [0.0] * 6
    t00 = time.perf_counter()
    for i in range(N):
        t = time.perf_counter()
        if i >= 2: sA.wait_event(done[(i - 2) % 4])
        acc[0] += time.perf_counter() - t; t = time.perf_counter()
        with torch.cuda.stream(sA):
            for _ in range(heavy): x.add_(1.0)
        acc[1] += time.perf_counter() - t; t = time.perf_counter()
        ready.record(sA)
        acc[2] += time.perf_counter() - t; t = time.perf_counter()
        cS.wait_event(ready)
        acc[3] += time.perf_counter() - t; t = time.perf_counter()
        with torch.cuda.stream(cS):
            if use_copy: out.copy_(counts, non_blocking=True)
            if use_kernel_on_c: out.add_(1)
        acc[4] += time.perf_counter() - t; t = time.perf_counter()
        done[i % 4].record(cS)
        acc[5] += time.perf_counter() - t
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(name, "host/step %.1f us, total/step %.1f us;" % ((t1 - t00) / N * 1e6, (t2 - t00) / N * 1e6),
          "wait %.1f kern %.1f rec %.1f cwait %.1f cwork %.1f crec %.1f" % tuple(a / N * 1e6 for a in acc))
run("copy on c, 1 kernel ", True, False, 1)
run("kernel on c, 1 kernel", False, True, 1)
run("copy on c, 20 kernels", True, False, 20)
run("kernel on c, 20 kern ", False, True, 20)
run("nothing on c, 20 kern", False, False, 20)
