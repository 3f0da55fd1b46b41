"""The package's synthetic-workload generator (efficientvlm_amd/workload.py: what bench.py feeds the trainer) against
oracle/synth.py, the generator the golden fixtures were captured with: same geometry tables, byte-identical batches."""
import torch

from efficientvlm_amd import workload as W
from oracle import synth


def test_geometries_and_batches_are_identical():
    assert W.GEOMS == synth.GEOMS
    for name, B, ragged in (("tiny", 5, True), ("tiny", 4, False), ("full", 3, True)):
        a = W.make_batch(W.GEOMS[name], B, seed=11, ragged=ragged)
        b = synth.make_batch(synth.GEOMS[name], B, seed=11, ragged=ragged)
        assert a.keys() == b.keys()
        for k in a:
            assert a[k].dtype == b[k].dtype and torch.equal(a[k], b[k]), k


def test_model_config_carries_the_reference_keys():
    cfg = W.model_config(W.GEOMS["full"], "s")
    assert cfg["vision_config"]["num_hidden_layers"] == 6 and cfg["text_num_hidden_layers"] == 6
    assert W.model_config(W.GEOMS["full"], "t")["vision_config"]["num_hidden_layers"] == 12
    for k in ("use_clip_vit", "image_res", "patch_size", "text_encoder", "embed_dim", "temp"):
        assert k in cfg


def test_gd_loss_mix_is_the_reference_linear_combination():
    """distill.gd_loss_mix stacks the step's ~17 device scalars and forms every output as one weighted sum: same values
    and the same gradient per term as the expression of GeneralDistill.py:369-376 / :252-260 (general and region steps)"""
    import torch
    from efficientvlm_amd import distill
    g = torch.Generator().manual_seed(0)
    mk = lambda: torch.rand((), generator=g).requires_grad_(True)
    loss = {k: mk() for k in ("loss_itc", "loss_itm", "loss_mlm", "loss_bbox", "loss_giou")}
    kd = {k: mk() for k in ("text_attn", "text_hidden", "image_attn", "image_hidden", "itm_neg_attn", "itm_neg_hidden",
                            "itm_pos_attn", "itm_pos_hidden", "mlm_attn", "mlm_hidden", "itm_logits", "mlm_logits")}

    def ref(loss, kd):
        small = loss["loss_itc"] + loss["loss_itm"] + loss["loss_mlm"]
        if "loss_bbox" in loss:
            small = small + loss["loss_bbox"] + loss["loss_giou"]
        t = kd["text_attn"] + kd["text_hidden"]
        i = kd["image_attn"] + 0.1 * kd["image_hidden"]
        c = (kd["itm_neg_attn"] + kd["itm_neg_hidden"] + kd["itm_pos_attn"] + kd["itm_pos_hidden"] + kd["mlm_attn"] + kd["mlm_hidden"])
        k = kd["itm_logits"] + kd["mlm_logits"] + t + i + c
        return small * 0.6 + k * 0.4, dict(loss_small=small, loss_text_kd=t, loss_img_kd=i, loss_cross_kd=c, loss_kd=k)

    for L in (loss, {k: v for k, v in loss.items() if k in ("loss_itc", "loss_itm", "loss_mlm")}):
        a, ma = distill.gd_loss_mix(L, kd)
        b, mb = ref(L, kd)
        assert abs(float(a.detach()) - float(b.detach())) < 1e-6
        for k in mb:
            assert abs(float(ma[k]) - float(mb[k].detach())) < 1e-6, k
        leaves = list(L.values()) + list(kd.values())
        for x, y in zip(torch.autograd.grad(a, leaves), torch.autograd.grad(b, leaves)):
            assert abs(float(x) - float(y)) < 1e-7


def test_assign_state_partitions_the_gradient_slabs():
    """optim.FlatAdamW.assign_state: the Linear weights' ranges (left out of the step's zero-fill: their first dY^T X
    product writes them) and the ranges zero_grad(skip_assigned=True) still fills cover every slab element exactly once;
    embeddings (tied to a decoder or not), biases, LayerNorms and 1-D parameters are never in the skip set"""
    import torch
    from efficientvlm_amd.optim import FlatAdamW

    class M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.emb = torch.nn.Embedding(100, 64)
            self.a = torch.nn.Linear(64, 128)
            self.norm = torch.nn.LayerNorm(128)
            self.b = torch.nn.Linear(128, 64)
            self.dec = torch.nn.Linear(64, 100, bias=False)
            self.dec.weight = self.emb.weight              # tied
            self.small = torch.nn.Linear(8, 8)             # below the size floor
            self.scale = torch.nn.Parameter(torch.ones(()))
    m = M()
    opt = FlatAdamW(m, lr=1e-3)
    st = opt.assign_state(m)
    names = {p.grad.data_ptr(): n for n, p in m.named_parameters()}
    assert sorted(names[k] for k in st["skip"]) == ["a.weight", "b.weight"]
    for g in opt.flat_grads:
        g.fill_(7.0)
    opt.zero_grad(skip_assigned=True)
    for v in st["skip"].values():
        assert bool((v == 7.0).all())
        v.zero_()
    assert all(bool((g == 0).all()) for g in opt.flat_grads)
    opt.zero_grad()                                        # the plain form fills everything
    assert all(bool((g == 0).all()) for g in opt.flat_grads)
