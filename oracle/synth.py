"""Synthetic batches of the GD / ITR tensor contract.  TEST + BENCH INPUTS.

Shapes follow dataset/pretrain_dataset.py:233-281 (image, text_ids, text_atts, text_ids_masked,
masked_pos, masked_ids); values are seeded CPU draws so the reference run, the oracle and the HIP
path see byte-identical inputs.
"""
import torch

GEOMS = {
    # hidden 64 / 4 heads (d_h = 16) / 5 image tokens / 8 text tokens; 6+6 student, 12+12 teacher
    "tiny": dict(hidden=64, heads=4, ffn=128, vocab=128, max_pos=64, image_res=32, embed_dim=16,
                 L=8, M=3, s_vit_layers=6, t_vit_layers=12, s_text_layers=6, t_text_layers=12,
                 cls=1, sep=2, mask=3, lo=10, std=0.15),
    # X-VLM-small student / X-VLM-base teacher, 224^2, 30 tokens, 8 masked positions
    "full": dict(hidden=768, heads=12, ffn=3072, vocab=30522, max_pos=512, image_res=224, embed_dim=256,
                 L=30, M=8, s_vit_layers=6, t_vit_layers=12, s_text_layers=6, t_text_layers=12,
                 cls=101, sep=102, mask=103, lo=1000, std=0.02),
}


def make_batch(geom, B, seed, ragged=False, image_res=None):
    g = torch.Generator().manual_seed(seed)
    R = image_res or geom["image_res"]
    L, M = geom["L"], geom["M"]
    image = torch.randn(B, 3, R, R, generator=g)
    ids = torch.randint(geom["lo"], geom["vocab"], (B, L), generator=g)
    atts = torch.ones(B, L, dtype=torch.long)
    lens = [L] * B
    if ragged:
        for b in range(1, B, 2):  # odd rows are shorter and 0-padded
            lens[b] = max(M + 2, L - 2 - (b % 3))
    for b in range(B):
        ids[b, 0] = geom["cls"]
        ids[b, lens[b] - 1] = geom["sep"]
        ids[b, lens[b]:] = 0
        atts[b, lens[b]:] = 0
    ids_masked = ids.clone()
    masked_pos = torch.zeros(B, M, dtype=torch.long)
    masked_ids = torch.full((B, M), -100, dtype=torch.long)
    for b in range(B):
        n_mask = M if b % 2 == 0 else M - 1       # one padded slot (pos 0 / label -100) on odd rows
        perm = torch.randperm(lens[b] - 2, generator=g)[:n_mask] + 1
        perm, _ = torch.sort(perm)
        masked_pos[b, :n_mask] = perm
        masked_ids[b, :n_mask] = ids[b, perm]
        ids_masked[b, perm] = geom["mask"]
    return dict(image=image, text_ids=ids, text_atts=atts, text_ids_masked=ids_masked,
                masked_pos=masked_pos, masked_ids=masked_ids)


def make_region_batch(geom, n_img, R, seed, ragged=True):
    """a REGION batch of the GD recipe (dataset/pretrain_dataset.py:405-526 collate contract): `n_img` images expanded
    to R (text, region) rows by idx_to_group_img (unsorted, as random.sample leaves it), per-row patch masks
    image_atts [R, 1+P*P] (cls always 1; whole-image rows all ones), target boxes (cx, cy, w, h) in [0,1] and the
    is_image flags that mask whole-image rows out of the box losses."""
    g = torch.Generator().manual_seed(seed + 4242)
    out = make_batch(geom, R, seed, ragged=ragged)
    out["image"] = torch.randn(n_img, 3, geom["image_res"], geom["image_res"], generator=g)
    P = geom["image_res"] // 16
    idx = torch.cat([torch.arange(n_img), torch.randint(0, n_img, (max(R - n_img, 0),), generator=g)])[:R]
    out["idx_to_group_img"] = idx[torch.randperm(R, generator=g)].long()
    atts = torch.zeros(R, 1 + P * P, dtype=torch.long)
    bbox = torch.zeros(R, 4)
    is_image = torch.zeros(R, dtype=torch.long)
    for r in range(R):
        if r % 4 == 1:                                   # the image's own caption: full attention, box = whole image
            is_image[r] = 1
            atts[r] = 1
            bbox[r] = torch.tensor([0.5, 0.5, 1.0, 1.0])
            continue
        x0, y0 = (int(v) for v in torch.randint(0, P, (2,), generator=g))
        x1 = int(torch.randint(x0 + 1, P + 1, (1,), generator=g))
        y1 = int(torch.randint(y0 + 1, P + 1, (1,), generator=g))
        atts[r, 0] = 1
        for i in range(y0, y1):
            atts[r, 1 + P * i + x0:1 + P * i + x1] = 1
        # the dataset's boxes are pixel-exact, its masks patch-rounded: keep the box a little inside the patch rectangle
        jit = torch.rand(4, generator=g) * 0.3 / P
        bx0, by0, bx1, by1 = x0 / P + jit[0], y0 / P + jit[1], x1 / P - jit[2], y1 / P - jit[3]
        bbox[r] = torch.stack([(bx0 + bx1) / 2, (by0 + by1) / 2, bx1 - bx0, by1 - by0])
    out.update(image_atts=atts, target_bbox=bbox, is_image=is_image)
    return out


def make_vqa_batch(geom, B, seed, La=6):
    """a VQA fine-tune batch (Eff_VQA.py:95-99 after tokenisation, dataset/vqa_dataset.py collate): B (image, question)
    pairs, k[b] candidate answers per question flattened to sum(k) answer rows with per-answer weights; answers are
    [CLS] tokens [SEP] 0-padded to La (pad_token_id = 0)."""
    g = torch.Generator().manual_seed(seed + 777)
    base = make_batch(geom, B, seed, ragged=True)
    k = [1 + (b * 2 + seed) % 3 for b in range(B)]
    n = sum(k)
    ids = torch.zeros(n, La, dtype=torch.long)
    atts = torch.zeros(n, La, dtype=torch.long)
    for r in range(n):
        ln = 3 + (r * 5 + seed) % (La - 2)                 # 3 .. La tokens incl. [CLS] and [SEP]
        ids[r, 0] = geom["cls"]
        ids[r, 1:ln - 1] = torch.randint(geom["lo"], geom["vocab"], (ln - 2,), generator=g)
        ids[r, ln - 1] = geom["sep"]
        atts[r, :ln] = 1
    weights = torch.rand(n, generator=g) * 0.8 + 0.2
    return dict(image=base["image"], question_ids=base["text_ids"], question_atts=base["text_atts"],
                answer_ids=ids, answer_atts=atts, k=torch.tensor(k, dtype=torch.long), weights=weights)
