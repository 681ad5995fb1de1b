"""Time the sequence-attention kernels at the fine-tune shapes (B=256, 8 heads, T=25; self 25 keys, cross 256 keys)."""
import os, sys, time, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dig_amd import _lib as L
dev = torch.device("cuda:0")
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6
B, H, T = 256, 8, 25
hk = H * 64
for name, Lk, causal in (("self", 25, 1), ("cross", 256, 0)):
    q = torch.randn(B * T, hk, device=dev).bfloat16(); kv = torch.randn(B * Lk, 2 * hk, device=dev).bfloat16()
    out = torch.empty_like(q); lse = torch.empty(B, H, T, device=dev); lens = torch.randint(1, 26, (B,), device=dev) if causal else None
    dq = torch.empty_like(q); dkv = torch.empty_like(kv); do = torch.randn_like(q)
    f = lambda: L.call("dig_seq_attn_fwd", L.ptr(q), hk, L.ptr(kv), 2 * hk, L.ptr(kv[:, hk:]), 2 * hk, L.ptr(out), hk, L.ptr(lse), B, H, T, Lk,
                       ctypes.c_float(0.125), causal, L.ptr(lens), L.stream())
    bw = lambda: L.call("dig_seq_attn_bwd", L.ptr(q), hk, L.ptr(kv), 2 * hk, L.ptr(kv[:, hk:]), 2 * hk, L.ptr(do), hk, L.ptr(lse), L.ptr(dq), hk,
                        L.ptr(dkv), 2 * hk, L.ptr(dkv[:, hk:]), 2 * hk, B, H, T, Lk, ctypes.c_float(0.125), causal, L.ptr(lens), L.stream())
    print(f"{name}: fwd {bench(f):.0f} us  bwd {bench(bw):.0f} us")
