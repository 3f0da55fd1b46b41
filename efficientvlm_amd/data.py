"""The MLM batch contract of the GD recipe: what ImageTextJsonDataset.preprocess / collate_fn hand the
training step for one caption - text_ids, text_atts, text_ids_masked, masked_pos, masked_ids (reference:
dataset/pretrain_dataset.py:233-281) - with its whole-word / skip-gram masking (TextMaskingGenerator, :46-137) and the
caption normalisation of dataset/utils.py:33-57.  Host-side Python by nature (tokens and Python's `random`), as in the
reference; the draws are made in the reference's order, so under the same `random.seed` the batches are identical
(tests/golden/mlm_batch.json, captured from the reference's own classes).

The tokenizer is any object with the four members the reference touches: get_vocab(), convert_tokens_to_ids(),
cls_token / sep_token / mask_token / pad_token_id (+ tokenize() for raw captions)."""
import copy
import random
import re

import torch
import torch.nn.functional as F

PAD_MASK = -100          # label of an unused mask slot (ignored by the MLM cross entropy)


def pre_caption(caption, max_words):
    """dataset/utils.py:33-57: lower-case, strip punctuation, squeeze blanks, truncate to max_words"""
    raw = caption
    caption = re.sub(r"([,.'!?\"()*#:;~])", " ", caption.lower())
    caption = caption.replace("-", " ").replace("/", " ").replace("<person>", "person")
    caption = re.sub(r"\s{2,}", " ", caption).rstrip("\n").strip(" ")
    words = caption.split(" ")
    if len(words) > max_words:
        caption = " ".join(words[:max_words])
    if not caption:
        raise ValueError(f"pre_caption yields invalid text (raw: {raw})")
    return caption


class TextMaskingGenerator:
    """dataset/pretrain_dataset.py:46-137 (BERT WordPiece branch; the RoBERTa branch marks word starts with 'Ġ')."""

    def __init__(self, tokenizer, mask_prob, mask_max, skipgram_prb=0.2, skipgram_size=3, mask_whole_word=True,
                 use_roberta=False):
        vocab = tokenizer.get_vocab()
        self.id2token = {i: w for w, i in vocab.items()}
        if sorted(self.id2token) != list(range(len(self.id2token))):
            raise ValueError("tokenizer ids must be 0 .. len(vocab) - 1")       # the reference asserts the same (:53-54)
        self.cls_token, self.mask_token = tokenizer.cls_token, tokenizer.mask_token
        self.mask_prob, self.mask_max = mask_prob, mask_max
        self.skipgram_prb, self.skipgram_size = skipgram_prb, skipgram_size
        self.mask_whole_word, self.use_roberta = mask_whole_word, use_roberta

    def _word_span(self, tokens, st, end):
        """grow [st, end) to whole words (:79-93)"""
        if self.use_roberta:
            while st > 1 and tokens[st][0] != "Ġ":
                st -= 1
            while end < len(tokens) and tokens[end][0] != "Ġ":
                end += 1
        else:
            while st >= 0 and tokens[st].startswith("##"):
                st -= 1
            while end < len(tokens) and tokens[end].startswith("##"):
                end += 1
        return st, end

    def __call__(self, tokens):
        """tokens: [CLS] ...; returns (tokens with masks applied - IN PLACE, as the reference -, masked positions)"""
        n_pred = min(self.mask_max, max(1, int(round(len(tokens) * self.mask_prob))))
        if tokens[0] != self.cls_token:
            raise AssertionError("the first token must be [CLS]")
        order = list(range(1, len(tokens)))
        random.shuffle(order)                                            # draw 1: candidate order (:70)
        last = max(order)
        chosen = set()
        for pos in order:
            if len(chosen) >= n_pred:
                break
            if pos in chosen:
                continue
            width = 1
            if self.skipgram_prb > 0 and self.skipgram_size >= 2 and random.random() < self.skipgram_prb:   # (:95)
                width = random.randint(2, self.skipgram_size)
            st, end = (self._word_span(tokens, pos, pos + width) if self.mask_whole_word else (pos, pos + width))
            for mp in range(st, end):
                if 0 < mp <= last:
                    chosen.add(mp)
                else:
                    break
        masked = list(chosen)
        if len(masked) > n_pred:
            random.shuffle(masked)
            masked = masked[:n_pred]
        for pos in masked:                                               # 80 % [MASK], 10 % random word, 10 % kept (:125-129)
            if random.random() < 0.8:
                tokens[pos] = self.mask_token
            elif random.random() < 0.5:
                tokens[pos] = self.id2token[random.randint(0, len(self.id2token) - 1)]
        return tokens, masked


class MLMBatcher:
    """preprocess + collate of ImageTextJsonDataset (dataset/pretrain_dataset.py:233-281) for the text side of a batch."""

    def __init__(self, tokenizer, max_tokens=40, max_masks=8, max_words=40, mask_prob=0.25, skipgram_prb=0.2,
                 skipgram_size=3, mask_whole_word=True, tokenized=False, add_eos=True):
        self.tok = tokenizer
        self.max_tokens, self.max_masks, self.max_words = max_tokens, max_masks, max_words
        self.tokenized, self.add_eos = tokenized, add_eos
        self.mask_generator = TextMaskingGenerator(tokenizer, mask_prob, max_masks, skipgram_prb, skipgram_size, mask_whole_word)

    def preprocess(self, text):
        """one caption -> (text_ids, text_atts, text_ids_masked, masked_pos, masked_ids), python lists padded to
        max_tokens / max_masks"""
        tokens = text.strip().split(" ") if self.tokenized else self.tok.tokenize(pre_caption(text, self.max_words))
        tokens = [self.tok.cls_token] + tokens[:self.max_tokens - 1]
        if self.add_eos:
            tokens = tokens[:self.max_tokens - 1] + [self.tok.sep_token]
        n = len(tokens)
        if n < 2:
            raise AssertionError("len(word tokens) < 2")
        ids = self.tok.convert_tokens_to_ids(tokens)
        masked_tokens, masked_pos = self.mask_generator(copy.deepcopy(tokens))
        ids_masked = self.tok.convert_tokens_to_ids(masked_tokens)
        masked_ids = [ids[p] for p in masked_pos]
        pad = [self.tok.pad_token_id] * (self.max_tokens - n)
        free = self.max_masks - len(masked_ids)
        return (ids + pad, [1] * n + [0] * len(pad), ids_masked + pad, masked_pos + [0] * free, masked_ids + [PAD_MASK] * free)

    def collate(self, samples):
        """list of preprocess() results -> dict of int64 tensors [B, max_tokens] / [B, max_masks] (collate_fn :269-281)"""
        cols = list(zip(*samples))
        names = ("text_ids", "text_atts", "text_ids_masked", "masked_pos", "masked_ids")
        return {n: torch.tensor(c, dtype=torch.long) for n, c in zip(names, cols)}

    def __call__(self, captions):
        return self.collate([self.preprocess(c) for c in captions])


# ---------------------------------------------------------------------------------------------------------------------
# Bucket padding for the fine-tune steps (round 6).  The reference tokenises a batch with padding='longest'
# (Eff_Retrieval.py:97, Eff_VQA.py:97-98) and a VQA batch carries a variable number of answers per question
# (dataset/vqa_dataset.py:101-116), so nearly every batch of an epoch has its own shape - and a captured step (hipGraph)
# its own graph.  Here a batch is padded on to one of a FEW shapes: text to the next length of a short ladder, answer rows
# to the next multiple of a block.  The step's arithmetic stays that of the 'longest'-padded batch:
#   * padded tokens are padding tokens (id pad, attention mask 0): as keys they get the -10000 mask, i.e. probability 0;
#   * a padded answer row has weight 0 (no LM loss), the pad id throughout (labels -100) and attends to the last question;
#   * the rows such tokens / answers occupy as QUERIES hold arbitrary values - the distillation terms skip them in their
#     kernels (`extents`: device int32 words with the REAL extents, ops.Ragged) and are rescaled from the padded to the real
#     denominator (`kd_corr`: padded / real element counts, distill.*_loss_mix).
# Both vectors travel inside the batch dict, so a teacher prefetch copies them into its static buffers like any tensor.
# ---------------------------------------------------------------------------------------------------------------------
TEXT_BUCKETS = (16, 24, 32, 40, 48, 64)
# VQA shapes are a product (question length x answer length x answer rows): coarser ladders keep the number of captured
# kinds small - the text side of a 480 x 480 step is a few per cent of its work, padding it costs little
QUESTION_BUCKETS = (16, 32, 48, 64)
ANSWER_TOKEN_BUCKETS = (8, 16, 32)
ANSWER_ROW_BLOCK = 128


def _bucket(n, ladder):
    for b in ladder:
        if n <= b:
            return b
    return (n + 7) // 8 * 8                     # beyond the ladder: the next multiple of 8 (a shape of its own)


def _pad_cols(t, width, value):
    return t if t.shape[1] == width else F.pad(t, (0, width - t.shape[1]), value=value)


def bucket_pad_itr(batch, buckets=TEXT_BUCKETS, pad_token_id=0):
    """batch: dict(image, text_ids [B, L], text_atts [B, L] [, idx]) as Eff_Retrieval.py:97-101 builds it -> the same batch
    with the text padded to the next bucket length, plus extents = int32 [L, 0, 0, 0] and kd_corr = f32 [L' / L, 1]"""
    L = int(batch["text_ids"].shape[1])
    Lp = _bucket(L, buckets)
    out = dict(batch)
    out["text_ids"] = _pad_cols(batch["text_ids"], Lp, pad_token_id)
    out["text_atts"] = _pad_cols(batch["text_atts"], Lp, 0)
    dev = batch["text_ids"].device
    out["extents"] = torch.tensor([L, 0, 0, 0], dtype=torch.int32, device=dev)
    out["kd_corr"] = torch.tensor([Lp / L, 1.0], dtype=torch.float32, device=dev)
    return out


def bucket_pad_vqa(batch, question_buckets=QUESTION_BUCKETS, answer_buckets=ANSWER_TOKEN_BUCKETS, row_block=ANSWER_ROW_BLOCK,
                   pad_token_id=0):
    """batch: dict(image, question_ids / question_atts [B, Lq], answer_ids / answer_atts [R, La], k [B], weights [R]) as
    Eff_VQA.py:92-103 builds it (R = sum k) -> question and answer tokens padded to their bucket lengths, the answer rows
    to the next multiple of `row_block` (weight 0, all-pad rows credited to the LAST question: k[-1] grows), plus
    extents = int32 [Lq, La, R, 0] and kd_corr = f32 [Lq' / Lq, (R' La') / (R La)]"""
    Lq, (R, La) = int(batch["question_ids"].shape[1]), (int(v) for v in batch["answer_ids"].shape)
    Lqp, Lap = _bucket(Lq, question_buckets), _bucket(La, answer_buckets)
    Rp = (R + row_block - 1) // row_block * row_block
    out = dict(batch)
    out["question_ids"] = _pad_cols(batch["question_ids"], Lqp, pad_token_id)
    out["question_atts"] = _pad_cols(batch["question_atts"], Lqp, 0)
    ids, atts = _pad_cols(batch["answer_ids"], Lap, pad_token_id), _pad_cols(batch["answer_atts"], Lap, 0)
    w = batch["weights"]
    k = torch.as_tensor(batch["k"]).clone()
    if Rp != R:
        ids = F.pad(ids, (0, 0, 0, Rp - R), value=pad_token_id)
        atts = F.pad(atts, (0, 0, 0, Rp - R), value=0)
        w = F.pad(w, (0, Rp - R), value=0.0)
        k[-1] += Rp - R
    out.update(answer_ids=ids, answer_atts=atts, weights=w, k=k)
    dev = batch["question_ids"].device
    out["extents"] = torch.tensor([Lq, La, R, 0], dtype=torch.int32, device=dev)
    out["kd_corr"] = torch.tensor([Lqp / Lq, (Rp * Lap) / (R * La)], dtype=torch.float32, device=dev)
    return out
