#!/usr/bin/env python3
"""Capture the MLM batch contract from the REFERENCE's own code (build container only: /root/reference must exist).
TextMaskingGenerator and ImageTextJsonDataset.preprocess / collate_fn of dataset/pretrain_dataset.py cannot be imported
(torchvision / PIL / HDFS reader at module level), so the two classes' code objects are ast-extracted and executed against
a synthetic WordPiece vocabulary; inputs and outputs go to tests/golden/mlm_batch.json.  Nothing of the reference is stored."""
import ast, copy, json, os, random, re, sys
from random import randint, shuffle
from random import random as rand

import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "mlm_batch.json")


def vocab():
    words = ["[PAD]", "[unused0]", "[unused1]", "[CLS]", "[SEP]", "[MASK]"]
    stems = ["a", "the", "dog", "cat", "man", "woman", "red", "blue", "car", "tree", "run", "sit", "on", "in", "with", "two",
             "play", "ball", "street", "table", "green", "small", "large", "white", "black", "person", "walk", "photo"]
    words += stems + ["##s", "##ing", "##ed", "##er", "##ly", "##ness", "##es"]
    return {w: i for i, w in enumerate(words)}


class Tok:
    cls_token, sep_token, mask_token, pad_token_id = "[CLS]", "[SEP]", "[MASK]", 0

    def __init__(self):
        self.v = vocab()

    def get_vocab(self):
        return dict(self.v)

    def convert_tokens_to_ids(self, toks):
        return [self.v[t] for t in toks]

    def tokenize(self, text):            # toy WordPiece: known stems + suffix pieces
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


def reference_classes():
    src = open(os.path.join(REF, "dataset", "pretrain_dataset.py")).read()
    tree = ast.parse(src)
    gen = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "TextMaskingGenerator")
    ds = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "ImageTextJsonDataset")
    keep = [n for n in ds.body if isinstance(n, ast.FunctionDef) and n.name in ("preprocess", "collate_fn")]
    shell = ast.ClassDef(name="RefDataset", bases=[], keywords=[], body=keep, decorator_list=[])
    usrc = open(os.path.join(REF, "dataset", "utils.py")).read()
    pc = next(n for n in ast.parse(usrc).body if isinstance(n, ast.FunctionDef) and n.name == "pre_caption")
    mod = ast.Module(body=[pc, gen, shell], type_ignores=[])
    ast.fix_missing_locations(mod)
    ns = dict(randint=randint, shuffle=shuffle, rand=rand, random=random, copy=copy, re=re, torch=torch, print=lambda *a, **k: None)
    exec(compile(mod, "<reference dataset classes>", "exec"), ns)
    return ns["TextMaskingGenerator"], ns["RefDataset"]


def main():
    Gen, Ref = reference_classes()
    tok = Tok()
    captions = ["A man walking two dogs on the street.", "the red car", "Two women playing with a small ball - in the green tree/table!",
                "cats sitting", "a person", "dogs " * 60, "The larger whiteness of the blackest trees runs smallly"]
    cases = []
    for cfg in (dict(max_tokens=40, max_masks=8, max_words=40, mask_prob=0.25, skipgram_prb=0.2, skipgram_size=3, mask_whole_word=True),
                dict(max_tokens=12, max_masks=4, max_words=10, mask_prob=0.5, skipgram_prb=0.6, skipgram_size=3, mask_whole_word=True),
                dict(max_tokens=30, max_masks=8, max_words=30, mask_prob=0.25, skipgram_prb=0.0, skipgram_size=3, mask_whole_word=False)):
        for seed in (0, 1, 2, 3, 4):
            ds = Ref.__new__(Ref)
            ds.tokenizer, ds.tokenized, ds.add_eos = tok, False, True
            ds.cls_token, ds.eos_token, ds.pad_token_id = tok.cls_token, tok.sep_token, tok.pad_token_id
            ds.max_words, ds.max_tokens, ds.max_masks, ds.PAD_mask = cfg["max_words"], cfg["max_tokens"], cfg["max_masks"], -100
            ds.mask_generator = Gen(tok, cfg["mask_prob"], cfg["max_masks"], cfg["skipgram_prb"], cfg["skipgram_size"], cfg["mask_whole_word"])
            random.seed(seed)
            samples = [ds.preprocess(c) for c in captions]
            batch = ds.collate_fn([(None,) + tuple(s) for s in samples])        # image slot None, as for a text-only check
            cases.append(dict(cfg=cfg, seed=seed, out=[b.tolist() for b in batch[1:]]))
    with open(OUT, "w") as f:
        json.dump(dict(vocab=vocab(), captions=captions, cases=cases), f)
    print("wrote", OUT, len(cases), "cases")


if __name__ == "__main__":
    main()
