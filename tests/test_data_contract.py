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


def test_bucket_padding_keeps_the_batch_and_records_its_real_extents():
    """data.bucket_pad_itr / bucket_pad_vqa (round 6; reference: Eff_Retrieval.py:97, Eff_VQA.py:97-98 `padding='longest'`,
    dataset/vqa_dataset.py:101-116 a variable number of answers): the real tokens / rows are untouched, the padding is pad ids
    with attention 0 (answer rows: weight 0, credited to the last question so that sum k = rows), the shapes come from short
    ladders, and `extents` / `kd_corr` hold the real extents and the padded / real denominators"""
    from efficientvlm_amd.data import bucket_pad_itr, bucket_pad_vqa, TEXT_BUCKETS
    g = torch.Generator().manual_seed(0)
    seen = set()
    for L in (5, 16, 17, 23, 40, 41, 70):
        ids = torch.randint(5, 100, (3, L), generator=g)
        atts = torch.ones(3, L, dtype=torch.long)
        atts[1, L - 2:] = 0
        b = bucket_pad_itr(dict(image=torch.zeros(3, 3, 2, 2), text_ids=ids, text_atts=atts, idx=torch.arange(3)))
        Lp = b["text_ids"].shape[1]
        seen.add(Lp)
        assert Lp >= L and Lp % 8 == 0 and (Lp in TEXT_BUCKETS or L > TEXT_BUCKETS[-1])
        assert torch.equal(b["text_ids"][:, :L], ids) and torch.equal(b["text_atts"][:, :L], atts)
        assert int(b["text_ids"][:, L:].abs().sum()) == 0 and int(b["text_atts"][:, L:].sum()) == 0
        assert b["extents"].dtype == torch.int32 and b["extents"].tolist() == [L, 0, 0, 0]
        assert abs(float(b["kd_corr"][0]) - Lp / L) < 1e-6 and float(b["kd_corr"][1]) == 1.0 and torch.equal(b["idx"], torch.arange(3))
    assert seen == {16, 24, 40, 48, 72}
    k = torch.tensor([2, 1, 4])
    R, La, Lq = int(k.sum()), 6, 11
    batch = dict(image=torch.zeros(3, 3, 2, 2), question_ids=torch.randint(5, 100, (3, Lq), generator=g),
                 question_atts=torch.ones(3, Lq, dtype=torch.long), answer_ids=torch.randint(5, 100, (R, La), generator=g),
                 answer_atts=torch.ones(R, La, dtype=torch.long), k=k, weights=torch.rand(R, generator=g) + 0.1)
    b = bucket_pad_vqa(batch, row_block=8)
    Rp, Lap, Lqp = b["answer_ids"].shape[0], b["answer_ids"].shape[1], b["question_ids"].shape[1]
    assert (Rp, Lap, Lqp) == (8, 8, 16) and int(b["k"].sum()) == Rp and b["k"].tolist() == [2, 1, 5] and k.tolist() == [2, 1, 4]
    assert torch.equal(b["answer_ids"][:R, :La], batch["answer_ids"]) and int(b["answer_ids"][R:].abs().sum()) == 0
    assert int(b["answer_atts"][R:].sum()) == 0 and int(b["answer_atts"][:, La:].sum()) == 0
    assert torch.equal(b["weights"][:R], batch["weights"]) and float(b["weights"][R:].abs().sum()) == 0.0
    assert b["extents"].tolist() == [Lq, La, R, 0]
    assert abs(float(b["kd_corr"][0]) - Lqp / Lq) < 1e-6 and abs(float(b["kd_corr"][1]) - (Rp * Lap) / (R * La)) < 1e-6
    # a batch that already sits on its buckets is returned with the same shapes
    b2 = bucket_pad_vqa(b, row_block=8)
    assert b2["answer_ids"].shape == b["answer_ids"].shape and b2["question_ids"].shape == b["question_ids"].shape
