"""Persistent forward tiles (bk 544 / 564) against the one-tile-per-workgroup kernels (bk 244 / 264): same arithmetic, so bit-identical."""
import sys, torch
sys.path.insert(0, ".")
from dig_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
ok = True
for (I, J, R, kw, a, b) in [(65536, 1536, 384, dict(act=1), 244, 544), (65536, 1536, 384, dict(act=1, pre=True), 244, 544), (65536, 1152, 384, dict(), 244, 544),
                            (65536, 384, 384, dict(resid=True), 264, 564), (65536, 384, 1536, dict(resid=True), 264, 564), (1000, 384, 384, dict(resid=True), 264, 564),
                            (65536 + 72, 1152, 384, dict(), 244, 544), (4096, 512, 512, dict(act=1), 244, 544), (32768, 1536, 384, dict(act=1, pre=True), 264, 564)]:
    x = torch.randn(I, R, device=dev).bfloat16(); w = (torch.randn(J, R, device=dev) * 0.05).bfloat16()
    bias = torch.randn(J, device=dev); res = torch.randn(I, J, device=dev).bfloat16()
    outs = []
    for bk in (a, b):
        y = torch.full((I, J), 7.0, device=dev, dtype=torch.bfloat16); pre = torch.full((I, J), 7.0, device=dev, dtype=torch.bfloat16)
        args = dict(out=y, bias=bias, bk=bk)
        if kw.get("pre"): args["pre"] = pre
        if kw.get("resid"): args["resid"] = res
        if kw.get("act"): args["act"] = kw["act"]
        for _ in range(3):
            ops.gemm(x, w, I, J, R, **args)
        outs.append((y, pre))
    same = torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    print(I, J, R, kw, "bit-identical" if same else "DIFFERENT", flush=True)
    ok &= same
print("ALL_OK" if ok else "SOME_FAIL")
