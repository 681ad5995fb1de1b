"""dig_amd.torch_ops: the C-ABI kernels as torch.library custom operators (`dig::linear`, `dig::layer_norm`, `dig::attention`,
`dig::mlp_block`).  CPU: schemas and fake (meta) implementations.  GPU: forward and autograd against fp32 torch references of the reference
modules' math (modeling_finetune.py: Mlp :53-60, Attention :87-125, Block :150-158), and torch.library.opcheck."""
import pytest
import torch
import torch.nn.functional as F

import dig_amd.torch_ops  # noqa: F401  (registers the operators)


def rel(a, b):
    return ((a.float() - b.float()).norm() / (b.float().norm() + 1e-20)).item()


def test_operators_are_registered_with_schemas_and_fake_kernels():
    sch = {n: str(getattr(torch.ops.dig, n).default._schema) for n in ("linear", "layer_norm", "attention", "mlp_block",
                                                                       "linear_bwd", "layer_norm_bwd", "attention_bwd", "mlp_block_bwd")}
    assert sch["linear"] == "dig::linear(Tensor x, Tensor w, Tensor? bias) -> Tensor"
    assert sch["attention"].startswith("dig::attention(Tensor qkv, SymInt n_img, SymInt heads)")
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode():
        x = torch.empty(512, 384, dtype=torch.bfloat16, device="cuda")
        w = torch.empty(1152, 384, dtype=torch.bfloat16, device="cuda")
        b = torch.empty(1152, device="cuda")
        qkv = torch.ops.dig.linear(x, w, b)
        assert qkv.shape == (512, 1152) and qkv.dtype == torch.bfloat16
        ctx, lse = torch.ops.dig.attention(qkv, 2, 6)
        assert ctx.shape == (512, 384) and lse.shape == (12, 256) and lse.dtype == torch.float32
        g, be = torch.empty(384, device="cuda"), torch.empty(384, device="cuda")
        y, mu, rs = torch.ops.dig.layer_norm(x, g, be, 1e-6)
        assert y.shape == x.shape and mu.shape == (512,)
        w1, b1 = torch.empty(1536, 384, dtype=torch.bfloat16, device="cuda"), torch.empty(1536, device="cuda")
        w2, b2 = torch.empty(384, 1536, dtype=torch.bfloat16, device="cuda"), torch.empty(384, device="cuda")
        out = torch.ops.dig.mlp_block(x, g, be, 1e-6, w1, b1, w2, b2)
        assert out[0].shape == (512, 384) and out[4].shape == (512, 1536)


@pytest.mark.gpu
def test_linear_layer_norm_attention_forward_and_autograd_vs_torch():
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(5)
    rn = lambda *s: torch.randn(*s, generator=g).to(dev)
    n_img, H, D = 4, 6, 384
    R = n_img * 256
    x = rn(R, D).bfloat16().requires_grad_(True)
    lw, lb = (1 + 0.2 * rn(D)).requires_grad_(True), (0.3 * rn(D)).requires_grad_(True)
    wq = (rn(3 * D, D) * 0.06).bfloat16().requires_grad_(True)
    bq = (rn(3 * D) * 0.2).requires_grad_(True)
    # --- operators
    ln, _, _ = torch.ops.dig.layer_norm(x, lw, lb, 1e-6)
    qkv = torch.ops.dig.linear(ln, wq, bq)
    ctx, _ = torch.ops.dig.attention(qkv, n_img, H)
    dctx = rn(R, D).bfloat16()
    ctx.backward(dctx)
    # --- fp32 torch reference of the same math (q is used as it comes out of the linear layer: the operator takes pre-scaled q)
    xr = x.detach().float().requires_grad_(True)
    lwr, lbr = lw.detach().clone().requires_grad_(True), lb.detach().clone().requires_grad_(True)
    wqr, bqr = wq.detach().float().requires_grad_(True), bq.detach().clone().requires_grad_(True)
    lnr = F.layer_norm(xr, (D,), lwr, lbr, 1e-6)
    qkvr = lnr @ wqr.t() + bqr
    q, k, v = (t.reshape(n_img, 256, H, 64).permute(0, 2, 1, 3) for t in qkvr.split(D, dim=1))
    ctxr = (torch.softmax(q @ k.transpose(-1, -2), -1) @ v).permute(0, 2, 1, 3).reshape(R, D)
    ctxr.backward(dctx.float())
    assert rel(ln, lnr) < 6e-3 and rel(qkv, qkvr) < 1e-2 and rel(ctx, ctxr) < 2e-2
    assert rel(x.grad, xr.grad) < 5e-2 and rel(wq.grad, wqr.grad) < 5e-2 and rel(bq.grad, bqr.grad) < 5e-2
    assert rel(lw.grad, lwr.grad) < 5e-2 and rel(lb.grad, lbr.grad) < 5e-2
    assert wq.grad.dtype == torch.bfloat16 and bq.grad.dtype == torch.float32


@pytest.mark.gpu
def test_mlp_block_forward_and_autograd_vs_torch():
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(6)
    rn = lambda *s: torch.randn(*s, generator=g).to(dev)
    R, D, Fh = 1000, 384, 1536
    x = (rn(R, D) * 1.3).bfloat16().requires_grad_(True)
    lw, lb = (1 + 0.2 * rn(D)).requires_grad_(True), (0.3 * rn(D)).requires_grad_(True)
    w1, b1 = (rn(Fh, D) * 0.06).bfloat16().requires_grad_(True), (rn(Fh) * 0.3).requires_grad_(True)
    w2, b2 = (rn(D, Fh) * 0.04).bfloat16().requires_grad_(True), (rn(D) * 0.3).requires_grad_(True)
    out = torch.ops.dig.mlp_block(x, lw, lb, 1e-6, w1, b1, w2, b2)[0]
    dout = rn(R, D).bfloat16()
    out.backward(dout)
    ps = [x, lw, lb, w1, b1, w2, b2]
    rs = [p.detach().float().requires_grad_(True) for p in ps]
    xr, lwr, lbr, w1r, b1r, w2r, b2r = rs
    outr = xr + F.linear(F.gelu(F.linear(F.layer_norm(xr, (D,), lwr, lbr, 1e-6), w1r, b1r)), w2r, b2r)
    outr.backward(dout.float())
    assert rel(out, outr) < 1e-2
    for name, p, r in zip(("x", "ln_w", "ln_b", "w1", "b1", "w2", "b2"), ps, rs):
        assert p.grad is not None and rel(p.grad, r.grad) < 5e-2, (name, rel(p.grad, r.grad))
    assert w1.grad.dtype == torch.bfloat16 and b1.grad.dtype == torch.float32


@pytest.mark.gpu
def test_opcheck_schema_fake_and_autograd_registration():
    dev = torch.device("cuda:0")
    x = torch.randn(512, 384, device=dev).bfloat16().requires_grad_(True)
    w = (torch.randn(1152, 384, device=dev) * 0.05).bfloat16().requires_grad_(True)
    b = torch.randn(1152, device=dev).requires_grad_(True)
    tests = ("test_schema", "test_faketensor", "test_autograd_registration")
    torch.library.opcheck(torch.ops.dig.linear.default, (x, w, b), test_utils=tests)
    qkv = (torch.randn(512, 1152, device=dev) * 0.5).bfloat16().requires_grad_(True)
    torch.library.opcheck(torch.ops.dig.attention.default, (qkv, 2, 6), test_utils=tests)
    g, be = torch.ones(384, device=dev, requires_grad=True), torch.zeros(384, device=dev, requires_grad=True)
    torch.library.opcheck(torch.ops.dig.layer_norm.default, (x, g, be, 1e-6), test_utils=tests)
