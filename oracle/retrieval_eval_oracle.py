"""CPU restatement of the retrieval evaluation / rerank loop, Eff_Retrieval.py:215-319 (TEST INFRASTRUCTURE: only tests/,
smoke() and bench.py's cpu_baseline leg may import this package).

Tensor-level: the reference's tokenizer / dataset plumbing is replaced by already-tokenised `text_ids`, `text_atts` and an
`images` tensor.  Built from the fixture-pinned forward functions of oracle/xvlm_oracle.py.  PINNED: the reference driver cannot be imported
(top-level ruamel / apex / dataset imports), but its `evaluation` function can be ast-extracted and run on the reference's
own model - oracle/gen_golden.py gen_rerank did, and tests/test_oracle_golden.py holds this restatement to those score
matrices (tests/golden/rerank_tiny.npz: one rank and both shards of a 2-rank run).
"""
import torch

from . import xvlm_oracle as O


@torch.no_grad()
def evaluation_scores(sd, cfg, images, text_ids, text_atts, k_test, zs=None, rank=0, world=1, text_bs=256):
    z = zs or {}
    bp = O._bert_prefix(sd)
    text_feats, text_embeds = [], []
    for i in range(0, text_ids.shape[0], text_bs):                     # :243-254
        tf = O.bert_model(sd, bp, cfg, input_ids=text_ids[i:i + text_bs], attention_mask=text_atts[i:i + text_bs], mode="text",
                          head_z=z.get("text_head_z"), mlp_z=z.get("text_intermediate_z"))[0]
        text_feats.append(tf)
        text_embeds.append(torch.nn.functional.normalize(
            torch.nn.functional.linear(tf[:, 0, :], sd["text_proj.weight"], sd["text_proj.bias"]), dim=-1))
    text_feats, text_embeds = torch.cat(text_feats), torch.cat(text_embeds)
    image_feats = O.vit_forward(sd, "vision_encoder.", images, cfg, head_z=z.get("vision_head_z"),
                                mlp_z=z.get("vision_intermediate_z"))[0]                         # :256-270
    image_embeds = torch.nn.functional.normalize(
        torch.nn.functional.linear(image_feats[:, 0, :], sd["vision_proj.weight"], sd["vision_proj.bias"]), dim=-1)
    sims = image_embeds @ text_embeds.t()                                                         # :272
    n_img, n_txt = sims.shape

    def itm_score(img, txt, atts):
        out = O.bert_model(sd, bp, cfg, encoder_embeds=txt, attention_mask=atts, encoder_hidden_states=img,
                           encoder_attention_mask=torch.ones(img.shape[:2], dtype=torch.long), mode="fusion",
                           head_z=z.get("cross_head_z"), mlp_z=z.get("cross_intermediate_z"))[0]
        return O.build_mlp_fwd(sd, "itm_head.", out[:, 0, :])[:, 1]

    i2t = torch.full((n_img, n_txt), -100.0)
    step = n_img // world + 1                                                                     # :277-279
    start, end = rank * step, min(n_img, rank * step + step)
    for i in range(start, end):                                                                   # :281-293
        _, idx = sims[i].topk(k=k_test, dim=0)
        i2t[i, idx] = itm_score(image_feats[i].repeat(k_test, 1, 1), text_feats[idx], text_atts[idx])
    t2i = torch.full((n_txt, n_img), -100.0)
    simt = sims.t()
    step = n_txt // world + 1                                                                     # :298-300
    start, end = rank * step, min(n_txt, rank * step + step)
    for i in range(start, end):                                                                   # :302-315
        _, idx = simt[i].topk(k=k_test, dim=0)
        t2i[i, idx] = itm_score(image_feats[idx], text_feats[i].repeat(k_test, 1, 1), text_atts[i].repeat(k_test, 1))
    return i2t, t2i          # distributed: the driver SUMs these across ranks (:317-319), -100 fill included
