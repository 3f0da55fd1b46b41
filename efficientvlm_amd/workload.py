"""The benchmark workload of BASELINE.json configs[0..1]: geometry of the X-VLM-small student / X-VLM-base teacher pair, the
config dict the reference constructors take, and the synthetic general batch of the GD tensor contract
(reference: dataset/pretrain_dataset.py:233-281 - image, text_ids, text_atts, text_ids_masked, masked_pos, masked_ids;
SURVEY.md 8d).  Values are seeded CPU draws, so every consumer (bench.py, the parity tests' oracle) sees byte-identical
inputs; tests/test_workload.py holds this generator to oracle/synth.py, which the golden fixtures were captured with."""
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


def model_config(geom, role, image_res=None, sparsity=0.25, dropout=0.0):
    """the config dict the reference constructors take (Pretrain_XVLM_small_4m.yaml keys), with the json files inlined"""
    vit_layers, text_layers = geom[f"{role}_vit_layers"], geom[f"{role}_text_layers"]
    vision = {"ckpt": "none", "vision_width": geom["hidden"], "patch_size": 16, "hidden_act": "quick_gelu",
              "num_attention_heads": geom["heads"], "attention_dropout": 0.0, "intermediate_size": geom["ffn"],
              "num_hidden_layers": vit_layers, "local_attn_depth": 2 if vit_layers == 6 else 4}
    bert = {"hidden_size": geom["hidden"], "num_attention_heads": geom["heads"], "intermediate_size": geom["ffn"],
            "num_hidden_layers": 12, "hidden_act": "gelu", "hidden_dropout_prob": dropout,
            "attention_probs_dropout_prob": dropout, "layer_norm_eps": 1e-12, "max_position_embeddings": geom["max_pos"],
            "type_vocab_size": 2, "vocab_size": geom["vocab"], "pad_token_id": 0, "initializer_range": 0.02}
    return {"use_clip_vit": True, "use_swin": False, "vision_config": vision, "image_res": image_res or geom["image_res"],
            "patch_size": 16, "text_encoder": bert, "text_num_hidden_layers": text_layers, "embed_dim": geom["embed_dim"],
            "temp": 0.07, "accelerator": {"FP16_OPT_LEVEL": "O0"}, "sparsity": sparsity, "load_params": False}
