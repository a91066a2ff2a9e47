"""CPU: tokenizer and memory-mapped token datasets against fixtures produced by the reference's own code
(tests/golden/gen_io_golden.py): the data formats either side of the hot path."""
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_tokenizer_matches_reference_ids_and_special_tokens():
    from emdr2_amd.tokenizer import BertWordPieceTokenizer
    ref = json.load(open(os.path.join(GOLD, "tokenizer_ref.json"), encoding="utf-8"))
    vocab = os.path.join(GOLD, "tokenizer_vocab.txt")
    for extra in (0, 100):
        t = BertWordPieceTokenizer(vocab, lower_case=True, vocab_extra_ids=extra)
        c = ref["cases"][str(extra)]
        assert (t.vocab_size, t.cls, t.sep, t.pad, t.mask, t.bos_token_id, t.eos_token_id) == \
               (c["vocab_size"], c["cls"], c["sep"], c["pad"], c["mask"], c["bos"], c["eos"])
        assert t.vocab.get("<extra_id_0>") == c["extra_first"] and t.vocab.get("<extra_id_99>") == c["extra_last"]
        for s, ids, dec in zip(ref["strings"], c["ids"], c["decoded"]):
            assert t.tokenize(s) == ids, s
            assert t.decode(ids) == dec, s
    tc = BertWordPieceTokenizer(vocab, lower_case=False)
    for s, ids in zip(ref["strings"], ref["cased_ids"]):
        assert tc.tokenize(s) == ids, s


def test_vocab_padding_rule():
    from emdr2_amd.tokenizer import vocab_size_with_padding
    assert vocab_size_with_padding(30524) == 30592 and vocab_size_with_padding(30624) == 30720 and vocab_size_with_padding(128) == 128


def test_mmap_dataset_reads_reference_files_and_round_trips(tmp_path):
    from emdr2_amd.data.indexed_dataset import MMapIndexedDataset, MMapIndexedDatasetBuilder, make_dataset
    ref = json.load(open(os.path.join(GOLD, "mmap_ref.json")))
    d = make_dataset(os.path.join(GOLD, "mmap_ref"))
    assert len(d) == len(ref["seqs"]) and str(d.dtype) == ref["dtype"]
    assert [x.tolist() for x in d[0:len(d)]] == ref["seqs"]
    assert d.doc_idx.tolist() == ref["doc_idx"]
    assert d.get(4, offset=10, length=5).tolist() == ref["seqs"][4][10:15]
    flat, off = d.flat_tokens()
    assert flat.tolist() == [t for s in ref["seqs"] for t in s] and off.tolist() == np.cumsum([0] + [len(s) for s in ref["seqs"]]).tolist()
    # our writer produces byte-identical files
    b = MMapIndexedDatasetBuilder(str(tmp_path / "x.bin"), dtype=d.dtype)
    for i, s in enumerate(ref["seqs"]):
        b.add_item(s)
        if i % 2 == 1:
            b.end_document()
    b.finalize(str(tmp_path / "x.idx"))
    for ext in (".bin", ".idx"):
        assert open(str(tmp_path / "x") + ext, "rb").read() == open(os.path.join(GOLD, "mmap_ref") + ext, "rb").read()
    with pytest.raises(ValueError):
        open(str(tmp_path / "bad.idx"), "wb").write(b"nonsense-nonsense-nonsense")
        open(str(tmp_path / "bad.bin"), "wb").write(b"")
        MMapIndexedDataset(str(tmp_path / "bad"))


def test_answer_presence_validation_matches_the_reference():
    from emdr2_amd.tasks.openqa.dense_retriever.evaluation.qa_validation import calculate_matches, has_answer
    ref = json.load(open(os.path.join(GOLD, "retrieval_ref.json")))
    for answers, text, match, expect in ref["cases"]:
        assert has_answer(answers, text, match) == expect, (answers, text, match)
    docs = {int(k): tuple(v) for k, v in ref["docs"].items()}
    stats = calculate_matches(docs, [q[0] for q in ref["questions"]], [(q[1], q[2]) for q in ref["questions"]], match_type="string")
    assert list(stats.top_k_hits) == ref["top_k_hits"] and [list(h) for h in stats.questions_doc_hits] == ref["questions_doc_hits"]


def test_decode_maps_ids_of_the_padded_vocabulary_tail_to_unk():
    from emdr2_amd.tokenizer import BertWordPieceTokenizer
    t = BertWordPieceTokenizer(os.path.join(GOLD, "tokenizer_vocab.txt"), vocab_extra_ids=100)
    ids = t.tokenize("the capital of france")
    assert t.decode(ids + [t.vocab_size + 5]) == "the capital of france [UNK]"
