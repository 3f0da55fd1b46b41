"""State-dict schema (checkpoint ABI) of the X-VLM models on the hot path.

TEST INFRASTRUCTURE.  Lists every floating tensor name + shape the reference's modules register,
so tests can (a) regenerate the deterministic weights without the reference and (b) assert that
the drop-in modules expose exactly the reference's keys.
"""


def _lin(d, p, out_f, in_f):
    d[p + ".weight"] = (out_f, in_f)
    d[p + ".bias"] = (out_f,)


def _ln(d, p, n):
    d[p + ".weight"] = (n,)
    d[p + ".bias"] = (n,)


def vit_schema(cfg, p="vision_encoder."):
    """efficient_models/eff_vit.py:387-405,96-99,226-229,211-212"""
    h, f = cfg["hidden"], cfg["ffn"]
    n = (cfg["image_res"] // cfg["patch"]) ** 2 + 1
    d = {p + "class_embedding": (h,), p + "patch_embed.weight": (h, 3, cfg["patch"], cfg["patch"]),
         p + "pos_embed.weight": (n, h)}
    _ln(d, p + "pre_layrnorm", h)
    for i in range(cfg["vit_layers"]):
        lp = f"{p}encoder.layers.{i}."
        for nm in ("k_proj", "v_proj", "q_proj", "out_proj"):
            _lin(d, lp + "self_attn." + nm, h, h)
        _ln(d, lp + "layer_norm1", h)
        _lin(d, lp + "mlp.fc1", f, h)
        _lin(d, lp + "mlp.fc2", h, f)
        _ln(d, lp + "layer_norm2", h)
    _ln(d, p + "post_layernorm", h)
    return d


def bert_schema(cfg, p, max_pos):
    """efficient_models/eff_bert.py:171-183,234-240,370-371,439,454-455"""
    h, f = cfg["hidden"], cfg["ffn"]
    d = {p + "embeddings.word_embeddings.weight": (cfg["vocab"], h),
         p + "embeddings.position_embeddings.weight": (max_pos, h),
         p + "embeddings.token_type_embeddings.weight": (2, h)}
    _ln(d, p + "embeddings.LayerNorm", h)
    for i in range(cfg["text_layers"]):
        lp = f"{p}encoder.layer.{i}."
        blocks = ["attention"] + (["crossattention"] if i >= cfg["fusion_layer"] else [])
        for b in blocks:
            for nm in ("query", "key", "value"):
                _lin(d, f"{lp}{b}.self.{nm}", h, h)
            _lin(d, f"{lp}{b}.output.dense", h, h)
            _ln(d, f"{lp}{b}.output.LayerNorm", h)
        _lin(d, lp + "intermediate.dense", f, h)
        _lin(d, lp + "output.dense", h, f)
        _ln(d, lp + "output.LayerNorm", h)
    return d


def xvlm_schema(cfg, max_pos, mlm=True, bbox=True, l0=False):
    """efficient_models/xvlm.py:211-260 (+ model_pretrain / model_retrieval constructors)"""
    h, e = cfg["hidden"], cfg["embed_dim"]
    d = vit_schema(cfg)
    if mlm:
        d.update(bert_schema(cfg, "text_encoder.bert.", max_pos))
        cp = "text_encoder.cls.predictions."
        d[cp + "bias"] = (cfg["vocab"],)
        _lin(d, cp + "transform.dense", h, h)
        _ln(d, cp + "transform.LayerNorm", h)
        d[cp + "decoder.weight"] = (cfg["vocab"], h)      # tied to word_embeddings
        d[cp + "decoder.bias"] = (cfg["vocab"],)          # tied to cls.predictions.bias
    else:
        d.update(bert_schema(cfg, "text_encoder.", max_pos))
    _lin(d, "vision_proj", e, h)
    _lin(d, "text_proj", e, h)
    d["temp"] = ()
    heads = [("itm_head", 2)] + ([("bbox_head", 4)] if bbox else [])
    for nm, out in heads:
        _lin(d, nm + ".0", 2 * h, h)
        _ln(d, nm + ".1", 2 * h)
        _lin(d, nm + ".3", out, 2 * h)
    if l0:
        nv, nt = cfg["vit_layers"], cfg["fusion_layer"]
        nc = cfg["text_layers"] - nt
        H, f = cfg["heads"], cfg["ffn"]
        d.update({"l0_module.vision_head_loga": (nv, H), "l0_module.text_head_loga": (nt, H),
                  "l0_module.cross_head_loga": (2 * nc, H), "l0_module.vision_int_loga": (nv, f),
                  "l0_module.text_int_loga": (nt, f), "l0_module.cross_int_loga": (nc, f),
                  "l0_module.lambda_1": (), "l0_module.lambda_2": ()})
    return d


def vqa_schema(cfg, max_pos, l0=False):
    """efficient_models/model_generation.py:23-55 EffXVLMForVQA / models/model_generation.py:228-255 XVLMForVQA: image encoder,
    BertModel text encoder (no MLM head, no projection / matching heads), BertLMHeadModel decoder of
    num_dec_layers = number of fusion layers whose EVERY layer cross-attends (fusion_layer 0, encoder_width = hidden)"""
    h = cfg["hidden"]
    d = vit_schema(cfg)
    d.update(bert_schema(cfg, "text_encoder.", max_pos))
    nd = cfg["text_layers"] - cfg["fusion_layer"]
    d.update(bert_schema(dict(cfg, text_layers=nd, fusion_layer=0), "text_decoder.bert.", max_pos))
    cp = "text_decoder.cls.predictions."
    d[cp + "bias"] = (cfg["vocab"],)
    _lin(d, cp + "transform.dense", h, h)
    _ln(d, cp + "transform.LayerNorm", h)
    d[cp + "decoder.weight"] = (cfg["vocab"], h)          # tied to the decoder's word_embeddings
    d[cp + "decoder.bias"] = (cfg["vocab"],)
    if l0:
        nv, nt = cfg["vit_layers"], cfg["fusion_layer"]
        nc = cfg["text_layers"] - nt
        H, f = cfg["heads"], cfg["ffn"]
        d.update({"l0_module.vision_head_loga": (nv, H), "l0_module.text_head_loga": (nt, H),
                  "l0_module.cross_head_loga": (2 * nc, H), "l0_module.decoder_head_loga": (2 * nd, H),
                  "l0_module.vision_int_loga": (nv, f), "l0_module.text_int_loga": (nt, f),
                  "l0_module.cross_int_loga": (nc, f), "l0_module.decoder_int_loga": (nd, f),
                  "l0_module.lambda_1": (), "l0_module.lambda_2": ()})
    return d


def det_weights(schema, seed, std):
    from .detinit import det_tensor
    return {k: det_tensor(k, shp, seed, std) for k, shp in schema.items()}
