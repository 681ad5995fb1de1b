"""How two HIP streams of different priority share the chip (gpurun -- 'python tools/gpu_priority_probe.py').

A chain of N small dependent kernels (each far too small to fill the GPU) is queued on stream A, one large kernel on stream B; both are
queued before either can start (a host-released gate kernel holds them back).  Reported: when B's kernel starts and ends relative to A's
chain, for (A high, B normal), (A normal, B high) and equal priorities -- i.e. whether a lower-priority queue is served while a higher
one has back-to-back work, which is what decides if small kernels can be hidden beside a big one by putting them on another stream."""
import sys
import time
import torch

dev = torch.device("cuda:0")
lo_p, hi_p = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)


def run(pa, pb, n_small, order):
    A, B = torch.cuda.Stream(dev, priority=pa), torch.cuda.Stream(dev, priority=pb)
    small = torch.zeros(1 << 18, device=dev)                 # ~5-8 us per add_
    big = torch.zeros(1 << 28, device=dev)                   # 1 GiB read+write: ~300 us alone
    gate = torch.zeros(1 << 29, device=dev)
    torch.cuda.synchronize()
    ev = {k: torch.cuda.Event(enable_timing=True) for k in ("t0", "a0", "a1", "b0", "b1")}
    cur = torch.cuda.current_stream(dev)
    ev["t0"].record(cur)
    for _ in range(8):
        gate.add_(1.0)                                        # ~2.5 ms of work on the caller's stream: both chains are fully queued behind it
    A.wait_stream(cur); B.wait_stream(cur)

    def qa():
        with torch.cuda.stream(A):
            ev["a0"].record(A)
            for _ in range(n_small):
                small.add_(1.0)
            ev["a1"].record(A)

    def qb():
        with torch.cuda.stream(B):
            ev["b0"].record(B)
            big.add_(1.0)
            ev["b1"].record(B)
    (qa(), qb()) if order == "ab" else (qb(), qa())
    torch.cuda.synchronize()
    t = {k: ev["t0"].elapsed_time(v) * 1e3 for k, v in ev.items()}
    base = min(t["a0"], t["b0"])
    return {k: round(v - base, 1) for k, v in t.items() if k != "t0"}


for n in (40, 150, 400):
    for name, pa, pb in (("A high / B normal", hi_p, lo_p), ("A normal / B high", lo_p, hi_p), ("equal", lo_p, lo_p)):
        for order in ("ab", "ba"):
            r = run(pa, pb, n, order)
            print(f"{n:4d} small kernels on A, {name:18s} queued {order}: A chain {r['a0']:8.1f} .. {r['a1']:8.1f} us   B big kernel {r['b0']:8.1f} .. {r['b1']:8.1f} us", flush=True)
