"""Kernel dispatch rate of N concurrent streams (hipGraph replays of 45 tiny kernels each)."""
import sys, threading, time
import torch

dev = torch.device("cuda:0")
N = 15
streams = [torch.cuda.Stream(dev) for _ in range(N)]
xs = [torch.zeros(64, device=dev) for _ in range(N)]
graphs = []
for i in range(N):
    with torch.cuda.stream(streams[i]):
        for _ in range(3):
            xs[i].add_(1.0)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=streams[i]):
        for _ in range(45):
            xs[i].add_(1.0)
    graphs.append(g)
torch.cuda.synchronize()
for n in (1, 2, 4, 8, 15):
    bar = threading.Barrier(n + 1)
    R = 300
    def work(i):
        with torch.cuda.stream(streams[i]):
            bar.wait()
            for _ in range(R):
                graphs[i].replay()
            streams[i].synchronize()
        bar.wait()
    th = [threading.Thread(target=work, args=(i,)) for i in range(n)]
    [t.start() for t in th]
    bar.wait(); t0 = time.perf_counter(); bar.wait(); dt = time.perf_counter() - t0
    [t.join() for t in th]
    print(f"streams={n:2d}: {n * R * 45 / dt / 1e3:8.1f} k kernels/s  ({1e6 * dt / (R * 45):.2f} us per kernel per stream)", flush=True)
