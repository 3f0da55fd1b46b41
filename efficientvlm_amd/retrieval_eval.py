"""Retrieval evaluation / rerank loop - the tensor-level body of Eff_Retrieval.py:215-319: features of
every text and image, the ITC similarity matrix, and for each query the k_test best candidates re-scored by the fusion
layers + ITM head; each rank scores its slice of the queries and the score matrices are SUM-all-reduced (with the
reference's -100 fill, so entries no rank computed come out as -100 * world, as upstream).

MI355X-first differences from the reference loop (same numbers):
  * queries are re-scored `query_bs` at a time instead of one by one (one query x 256 candidates is 7 680 token rows - far
    too little to fill 256 CUs);
  * image -> text: the query image is NOT repeated k_test times; its K / V projections are computed once per image and
    shared by its candidates through the cross-attention `kv_index` (encoder_batch_index).
"""
import torch
import torch.distributed as dist

from . import ops
from .efficient_models.xvlm import _matmul_nt, mlp_head_forward


@torch.no_grad()
def evaluation_scores(model, images, text_ids, text_atts, k_test=256, zs="auto", image_bs=64, text_bs=256, query_bs=8,
                      rank=None, world=None, reduce=True):
    """-> (score_matrix_i2t [n_img, n_txt], score_matrix_t2i [n_txt, n_img]) float32 device tensors"""
    model.eval()
    if rank is None:
        on = dist.is_available() and dist.is_initialized()
        rank, world = (dist.get_rank(), dist.get_world_size()) if on else (0, 1)
    if zs == "auto":
        zs = model.l0_module.forward(training=False) if getattr(model, "l0_module", None) is not None else None
    z = zs or {}
    dev = images.device
    core = model._text_core()
    text_feats, text_embeds = [], []
    for i in range(0, text_ids.shape[0], text_bs):
        tf = model.get_text_embeds(text_ids[i:i + text_bs], text_atts[i:i + text_bs], head_z=z.get("text_head_z"),
                                   mlp_z=z.get("text_intermediate_z"))
        text_feats.append(tf)
        text_embeds.append(model.get_features(text_embeds=tf))
    text_feats, text_embeds = torch.cat(text_feats), torch.cat(text_embeds)
    image_feats, image_embeds = [], []
    for i in range(0, images.shape[0], image_bs):
        f = model.get_vision_embeds(images[i:i + image_bs], head_z=z.get("vision_head_z"),
                                    mlp_z=z.get("vision_intermediate_z"))[0]
        image_feats.append(f)
        image_embeds.append(model.get_features(image_embeds=f))
    image_feats, image_embeds = torch.cat(image_feats), torch.cat(image_embeds)
    sims = _matmul_nt(image_embeds.float(), text_embeds.float())
    n_img, n_txt = sims.shape
    n_tok = image_feats.shape[1]

    def itm_score(img, img_index, txt, atts):
        enc_atts = torch.ones((txt.shape[0], n_tok), dtype=torch.long, device=dev)
        out = core(encoder_embeds=txt, attention_mask=atts, encoder_hidden_states=img, encoder_attention_mask=enc_atts,
                   encoder_batch_index=img_index, return_dict=True, mode="fusion", head_z=z.get("cross_head_z"),
                   mlp_z=z.get("cross_intermediate_z"))
        return mlp_head_forward(model.itm_head, out.last_hidden_state[:, 0, :])[:, 1].float()

    i2t = torch.full((n_img, n_txt), -100.0, device=dev)
    step = n_img // world + 1
    start, end = rank * step, min(n_img, rank * step + step)
    for q0 in range(start, end, query_bs):
        q1 = min(end, q0 + query_bs)
        idx = sims[q0:q1].topk(k=k_test, dim=1).indices                           # [q, k]
        flat = idx.reshape(-1)
        owner = torch.arange(q1 - q0, device=dev).repeat_interleave(k_test)
        score = itm_score(image_feats[q0:q1], owner, text_feats[flat], text_atts[flat])
        i2t[q0:q1].scatter_(1, idx, score.view(q1 - q0, k_test))
    t2i = torch.full((n_txt, n_img), -100.0, device=dev)
    simt = sims.t().contiguous()
    step = n_txt // world + 1
    start, end = rank * step, min(n_txt, rank * step + step)
    for q0 in range(start, end, query_bs):
        q1 = min(end, q0 + query_bs)
        idx = simt[q0:q1].topk(k=k_test, dim=1).indices
        flat = idx.reshape(-1)
        owner = torch.arange(q0, q1, device=dev).repeat_interleave(k_test)
        # candidate images: distinct ones once, shared through the index (a query's candidates may repeat across queries)
        uniq, inv = torch.unique(flat, return_inverse=True)
        score = itm_score(image_feats[uniq], inv, text_feats[owner], text_atts[owner])
        t2i[q0:q1].scatter_(1, idx, score.view(q1 - q0, k_test))
    if reduce and world > 1:
        dist.barrier()
        dist.all_reduce(i2t, op=dist.ReduceOp.SUM)
        dist.all_reduce(t2i, op=dist.ReduceOp.SUM)
    return i2t, t2i
