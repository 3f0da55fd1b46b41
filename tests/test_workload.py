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
