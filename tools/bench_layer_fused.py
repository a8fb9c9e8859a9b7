"""Forward + backward of the layer's chain from the output of v_proj to the clamped features, at B x N:
tensor-library ops around BKCoreFunction versus FusedBKLayerFunction (two library launches around the operator)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "project-resnet-bk-an-o-n-language-model-architecture_amd"))
import torch
if os.environ.get('BKCORE_DEV_LIB'):
    from bkcore import _lib
    _lib.LIB_PATH = os.path.abspath(os.environ['BKCORE_DEV_LIB'])
import bkcore

B, N = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1024, 32768)
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
v0 = torch.randn(B, N, generator=g, device=dev)
h0 = torch.full((1, N), -2.0, device=dev)
b = torch.ones(1, N - 1, device=dev)
z = torch.tensor(1j, dtype=torch.complex64, device=dev)
gf = torch.randn(B, N, 2, generator=g, device=dev)


def unfused(v):
    x = torch.nan_to_num(v, nan=0.0, posinf=5.0, neginf=-5.0)
    x = 5.0 * torch.tanh(x / 5.0)
    he = h0.expand(B, -1) + x
    f, G = bkcore.BKCoreFunction.apply(he.to(torch.bfloat16), b.expand(B, -1).to(torch.bfloat16),
                                       b.expand(B, -1).to(torch.bfloat16), z)
    return torch.clamp(f, -5.0, 5.0)


def fused(v):
    return bkcore.FusedBKLayerFunction.apply(v, h0, b, b, z, 5.0, 5.0)[0]


def timed(fn, reps=20):
    def step():
        v = v0.detach().requires_grad_(True)
        torch.autograd.backward((fn(v),), (gf,))
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        step()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


tu, tf = timed(unfused), timed(fused)
print(json.dumps({"B": B, "N": N, "unfused_fwd_bwd_ms": tu, "fused_fwd_bwd_ms": tf, "speedup": tu / tf}))

# output stage: residual + bk_scale * Linear(2 -> D)(features), forward + backward, n positions
for n_pos, D in ((8 * 256, 256), (64 * 2048, 256), (1024 * 1024, 256)):
    f0 = torch.randn(n_pos, 2, generator=g, device=dev)
    lin = torch.nn.Linear(2, D).to(dev)
    sc = torch.ones(D, device=dev, requires_grad=True)
    r0 = torch.randn(n_pos, D, generator=g, device=dev)
    go = torch.randn(n_pos, D, generator=g, device=dev)

    def plain(f, r):
        return r + sc * lin(f)

    def fused_out(f, r):
        return bkcore.FusedSpectralOutputFunction.apply(f, lin.weight, lin.bias, sc, r)

    def timed2(fn, reps=20):
        def step():
            f = f0.detach().requires_grad_(True)
            r = r0.detach().requires_grad_(True)
            lin.zero_grad(set_to_none=True); sc.grad = None
            fn(f, r).backward(go)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            step()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    tu, tf = timed2(plain), timed2(fused_out)
    print(json.dumps({"output_stage_positions": n_pos, "D": D, "unfused_fwd_bwd_ms": tu, "fused_fwd_bwd_ms": tf,
                      "speedup": tu / tf}))

