"""Batch layout of the QA task (reference: tasks/openqa/e2eqa/train_data_utils.py:27-81,128-150 and the collate of
train_e2eqa.py:51-68).  Host-side int64 list building, unchanged in meaning: question -> [CLS] q [SEP] pad (S_ret),
answer -> dec_ids = [BOS] a pad, labels = a [EOS] pad, loss_mask (L)."""
import csv
from ast import literal_eval
from collections import OrderedDict
from copy import deepcopy

import numpy as np
import torch


def build_tokens_types_paddings_from_ids(src_ids, answer_text_ids, max_seq_length, decoder_seq_length, cls_id, sep_id, pad_id, bos_id, eos_id):
    """The reference's item layout (tasks/openqa/e2eqa/train_data_utils.py:27-81; pinned by tests/golden/a1_ref.npz) as array fills:
    encoder row = [CLS] question [SEP] then padding, the question cut so that both markers fit; its token types are 0 on the tokens and --
    like the reference -- `pad_id` on the padding; decoder input = [BOS] answer, labels = answer [EOS], both cut to the decoder length and
    padded; loss_mask marks the decoder-input positions that hold a token.  Returns python lists and the encoder token count."""
    S, L = int(max_seq_length), int(decoder_seq_length)
    question = np.asarray(list(src_ids), dtype=np.int64)[:max(S - 2, 0)]
    n_enc = question.size + 2
    enc = np.full(max(S, n_enc), pad_id, dtype=np.int64)
    enc[0], enc[1:n_enc - 1], enc[n_enc - 1] = cls_id, question, sep_id
    types = np.where(np.arange(enc.size) < n_enc, 0, pad_id)

    answer = np.asarray(list(answer_text_ids), dtype=np.int64)[:max(L - 1, 0)]
    n_dec = answer.size + 1
    if n_dec > L:
        raise AssertionError("decoder_seq_length must hold at least [BOS]")
    dec_in = np.full(L, pad_id, dtype=np.int64)
    labels = np.full(L, pad_id, dtype=np.int64)
    dec_in[0], dec_in[1:n_dec] = bos_id, answer
    labels[:answer.size], labels[answer.size] = answer, eos_id
    loss_mask = (np.arange(L) < n_dec).astype(np.int64)
    return enc.tolist(), types.tolist(), n_enc, dec_in.tolist(), labels.tolist(), loss_mask.tolist()


def build_sample(query_uid, question_ids, answer_ids, seq_length_ret, decoder_seq_length, cls_id, sep_id, pad_id, bos_id, eos_id, reference=None):
    """One dataset item with the reference's 10 keys (train_data_utils.py:152-173)."""
    enc, types, n_enc, dec_in, dec_out, loss_mask = build_tokens_types_paddings_from_ids(
        question_ids, answer_ids, seq_length_ret, decoder_seq_length, cls_id, sep_id, pad_id, bos_id, eos_id)
    ids = np.array(enc, dtype=np.int64)
    mask2d = ((ids[None, :] >= 1) * (ids[:, None] >= 1)).astype(np.int64)        # make_attention_mask (mask_creation_utils.py:5-14)
    return OrderedDict(query_uid=query_uid, query_ids_bert=enc, query_types=types, query_mask_bert=mask2d, query_ids_t5=enc,
                       query_ids_t5_len=n_enc, dec_ids=dec_in, labels=dec_out, loss_mask=loss_mask, reference=reference)


def collate(batch_data):
    """CustomDataLoader._collate_fn (train_e2eqa.py:51-68)."""
    t = OrderedDict()
    for d in batch_data:
        for k, v in d.items():
            t.setdefault(k, []).append(v)
    assert len(t) == 10
    for k in ("query_uid", "query_ids_bert", "query_types", "query_ids_t5", "query_ids_t5_len", "dec_ids", "labels"):
        t[k] = torch.tensor(np.array(t[k]), dtype=torch.int64)
    t["query_mask_bert"] = torch.tensor(np.array(t["query_mask_bert"]), dtype=torch.int64)
    t["loss_mask"] = torch.tensor(np.array(t["loss_mask"]), dtype=torch.float32)
    return t


class OpenQADataset(torch.utils.data.Dataset):
    """tasks/openqa/e2eqa/train_data_utils.py:105-200: tab-separated `question <TAB> ["answer", ...]` files; uid = -(line number) so it can
    never collide with a 1-based evidence id; one answer is sampled per access with the dataset's own RandomState(seed)."""

    def __init__(self, task_name, dataset_name, datapaths, tokenizer, max_seq_length, decoder_seq_length, seed=1234):
        self.np_rng = np.random.RandomState(seed=seed)
        self.task_name, self.dataset_name, self.tokenizer = task_name, dataset_name, tokenizer
        self.max_seq_length, self.decoder_seq_length = max_seq_length, decoder_seq_length
        self.samples = []
        for datapath in datapaths:
            self.samples.extend(self.process_samples_from_single_path(datapath))

    def __len__(self):
        return len(self.samples)

    def __getitem__(self, idx):
        raw = self.samples[idx]
        answers = deepcopy(raw['answers'])
        self.np_rng.shuffle(answers)
        t = self.tokenizer
        return build_sample(raw['uid'], t.tokenize(raw['question']), t.tokenize(answers[0]), self.max_seq_length, self.decoder_seq_length,
                            t.cls, t.sep, t.pad, t.bos_token_id, t.eos_token_id, reference=raw['answers'])

    @staticmethod
    def process_samples_from_single_path(filename):
        samples = []
        with open(filename, 'r') as f:
            for total, row in enumerate(csv.reader(f, delimiter='\t'), start=1):
                samples.append({'uid': -total, 'question': row[0], 'answers': literal_eval(row[1])})   # the reference uses eval()
        return samples
