import sys, torch
sys.path.insert(0, "/root/repo")
from efficientvlm_amd import ops as o
DEV="cuda"
def run(L, B, H, scale_in):
    dh, d = 64, H*64
    g = torch.Generator().manual_seed(5)
    x0 = (torch.randn(B, L, 3*d, generator=g) * scale_in).to(torch.bfloat16).to(DEV)
    gO = torch.randn(B, L, d, generator=g).to(torch.bfloat16).to(DEV)
    res = {}
    for store in (True, False):
        o.ATTN_STORE_P = store
        x = x0.clone().requires_grad_(True)
        O, P = o.self_attention(x, H, dh, 0.125, want_probs=False)
        (O.float()*gO.float()).sum().backward()
        res[store] = x.grad.float()
    o.ATTN_STORE_P = False
    xr = x0.float().requires_grad_(True)
    sp = lambda t: t.reshape(B, L, H, dh).transpose(1, 2)
    q, k, v = sp(xr[..., :d]), sp(xr[..., d:2*d]), sp(xr[..., 2*d:])
    Pr = torch.softmax(q @ k.transpose(-1, -2) * 0.125, -1)
    Or = (Pr @ v).transpose(1, 2).reshape(B, L, d)
    (Or*gO.float()).sum().backward()
    l2 = lambda a, b: float((a.double()-b.double()).norm()/b.double().norm())
    for nm, sl in (("q", slice(0, d)), ("k", slice(d, 2*d)), ("v", slice(2*d, 3*d))):
        print(f"L={L:4d} in-scale {scale_in}: d{nm}: stored-map {l2(res[True][..., sl], xr.grad[..., sl]):.4f}   recomputing {l2(res[False][..., sl], xr.grad[..., sl]):.4f}", flush=True)
for L, B, H in ((197, 4, 12), (577, 2, 12), (901, 1, 12)):
    for sc in (0.5, 1.5):
        run(L, B, H, sc)
