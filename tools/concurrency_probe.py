"""How well do concurrent B=64 folds fill one MI355X?  N host threads, each with its own Engine and
HIP stream, run K train steps; prints aggregate steps/s and the host-side cost of one call for
(a) plain launches and (b) a hipGraph replay of the same step (scalars frozen — timing only).

    python tools/concurrency_probe.py [--batch 64] [--steps 200]
"""
import argparse
import sys
import threading
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import ctypes as C                                       # noqa: E402
from multimodalsignal_amd import _lib as L               # noqa: E402
from multimodalsignal_amd.runtime import Engine          # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--samples", type=int, default=3840)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--threads", type=int, nargs="+", default=[1, 2, 4, 8, 15])
    ap.add_argument("--xcds", type=int, default=0, help="pin stream i to XCDs [(i*xcds)%8, +xcds) (0 = unrestricted torch streams)")
    ap.add_argument("--mode", choices=["launch", "graph", "both"], default="both")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    nmax = max(a.threads)
    g = torch.Generator(device="cpu").manual_seed(0)
    engines, xs, ys, streams, graphs = [], [], [], [], []
    for i in range(nmax):
        e = Engine(6, 2, dev)
        for k, v in e.named_param_views().items():
            if v.numel():
                v.copy_(torch.randn(v.shape, generator=g) * 0.1)
        engines.append(e)
        xs.append(torch.randn(a.batch, 6, a.samples, generator=g).to(dev))
        ys.append(torch.randint(0, 2, (a.batch,), generator=g).to(dev))
        if a.xcds:
            h = C.c_void_p()
            L.check(L.lib().msig_fold_stream_create((i * a.xcds) % 8, a.xcds, C.byref(h)), "msig_fold_stream_create")
            streams.append(torch.cuda.ExternalStream(h.value, dev))
        else:
            streams.append(torch.cuda.Stream(dev))
    if a.xcds:                                             # where do the masked streams really run?
        ids = torch.full((512,), -1, dtype=torch.int32, device=dev)
        for i in range(min(nmax, 8)):
            L.check(L.lib().msig_probe_xcd_ids(ids.data_ptr(), 512, C.c_void_p(streams[i].cuda_stream)), "probe")
            streams[i].synchronize()
            print(f"stream {i}: ran on XCDs {sorted(set(ids.cpu().tolist()))}", flush=True)

    def step(i, s):
        engines[i].train_step(xs[i], ys[i], lr=1e-3, weight_decay=1e-4, step=s, dropout_p=0.5, seed=i)

    for i in range(nmax):                                  # warm up + capture
        with torch.cuda.stream(streams[i]):
            for s in range(1, 4):
                step(i, s)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=streams[i]):
            step(i, 4)
        graphs.append(gr)
    torch.cuda.synchronize()

    def run(n, use_graph):
        host = [0.0] * n
        bar = threading.Barrier(n + 1)

        def work(i):
            with torch.cuda.stream(streams[i]):
                bar.wait()
                t = 0.0
                for s in range(a.steps):
                    t0 = time.perf_counter()
                    if use_graph:
                        graphs[i].replay()
                    else:
                        step(i, 5 + s)
                    t += time.perf_counter() - t0
                host[i] = t / a.steps
                streams[i].synchronize()
            bar.wait()

        th = [threading.Thread(target=work, args=(i,)) for i in range(n)]
        for t in th:
            t.start()
        torch.cuda.synchronize()
        bar.wait()
        t0 = time.perf_counter()
        bar.wait()
        dt = time.perf_counter() - t0
        for t in th:
            t.join()
        return n * a.steps / dt, sum(host) / n

    for use_graph in {"launch": (False,), "graph": (True,), "both": (False, True)}[a.mode]:
        for n in a.threads:
            sps, host = run(n, use_graph)
            print(f"{'graph ' if use_graph else 'launch'} threads={n:2d}  {sps:8.1f} steps/s  "
                  f"{sps * a.batch / 1e3:7.1f} k windows/s  per-fold step {1e3 * n / sps:6.2f} ms  host/call {1e6 * host:7.1f} us",
                  flush=True)


if __name__ == "__main__":
    main()
