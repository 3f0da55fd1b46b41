"""SURVEY.md 8f-4, data contract: efficientvlm_amd.data (pre_caption, TextMaskingGenerator, MLMBatcher) against
tests/golden/mlm_batch.json - inputs and outputs of the REFERENCE's own TextMaskingGenerator / ImageTextJsonDataset.preprocess /
collate_fn (dataset/pretrain_dataset.py:46-137, :233-281), captured by oracle/gen_mlm_fixture.py under fixed `random` seeds.
Index tensors: bit-exact."""
import json
import os
import random

import torch

from efficientvlm_amd.data import PAD_MASK, MLMBatcher, pre_caption

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mlm_batch.json")


class Tok:
    cls_token, sep_token, mask_token, pad_token_id = "[CLS]", "[SEP]", "[MASK]", 0

    def __init__(self, v):
        self.v = v

    def get_vocab(self):
        return dict(self.v)

    def convert_tokens_to_ids(self, toks):
        return [self.v[t] for t in toks]

    def tokenize(self, text):
        out = []
        for w in text.split(" "):
            if w in self.v:
                out.append(w); continue
            for suf in ("ness", "ing", "ed", "er", "ly", "es", "s"):
                if w.endswith(suf) and w[:-len(suf)] in self.v:
                    out += [w[:-len(suf)], "##" + suf]; break
            else:
                out.append("a")
        return out


def test_mlm_batches_are_identical_to_the_reference_datasets():
    fx = json.load(open(GOLDEN))
    tok = Tok(fx["vocab"])
    names = ("text_ids", "text_atts", "text_ids_masked", "masked_pos", "masked_ids")
    n_masked = 0
    for case in fx["cases"]:
        c = case["cfg"]
        b = MLMBatcher(tok, max_tokens=c["max_tokens"], max_masks=c["max_masks"], max_words=c["max_words"],
                       mask_prob=c["mask_prob"], skipgram_prb=c["skipgram_prb"], skipgram_size=c["skipgram_size"],
                       mask_whole_word=c["mask_whole_word"])
        random.seed(case["seed"])
        got = b(fx["captions"])
        for name, want in zip(names, case["out"]):
            w = torch.tensor(want, dtype=torch.long)
            assert got[name].dtype == torch.long and torch.equal(got[name], w), (case["cfg"], case["seed"], name)
        # the contract the training step relies on: unused mask slots are (position 0, label -100); labels are the ORIGINAL ids
        ids, pos, lab = got["text_ids"], got["masked_pos"], got["masked_ids"]
        used = lab != PAD_MASK
        assert bool((pos[~used] == 0).all()) and bool((torch.gather(ids, 1, pos)[used] == lab[used]).all())
        assert bool((got["text_atts"].sum(1) >= 2).all()) and got["text_ids"].shape == (len(fx["captions"]), c["max_tokens"])
        n_masked += int(used.sum())
    assert n_masked > 100


def test_pre_caption_normalisation():
    assert pre_caption("A man, walking: two-dogs/cats!  <person> here.\n", 40) == "a man walking two dogs cats person here"
    assert pre_caption("one two three four", 2) == "one two"
