"""Fixture generator (build container only): runs the REFERENCE's tokenizer and indexed-dataset builder on inputs created here and
stores inputs + expected outputs.  Nothing of the reference is copied; /root/reference is imported read-only.

    python tests/golden/gen_io_golden.py

Writes tests/golden/tokenizer_vocab.txt (synthetic vocabulary made here), tokenizer_ref.json (strings -> ids, decoded strings, special
ids for vocab_extra_ids 0 and 100) and mmap_ref.{bin,idx} + mmap_ref.json (sequences written by the reference's MMapIndexedDataset builder).
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from _ref_import import install_import_shims  # noqa: E402

install_import_shims()
from megatron.tokenizer.tokenizer import _BertWordPieceTokenizer  # noqa: E402
from megatron.data import indexed_dataset as ref_ds  # noqa: E402

WORDS = ("the of and in to was is for on as with by that at from his he it an are which this were be or had first one their has new its "
         "who not but also after two they have been other when during all into there time may more years over only school city world "
         "emperor capital river paris france what year did war end where born wrote song play ing ed s er ly un re able tion ##ing ##ed ##s "
         "##er ##ly ##tion ##able ##a ##b ##c ##d ##e ##i ##n ##o ##r ##t ##u a b c d e f g h i j k l m n o p q r s t u v w x y z 0 1 2 3 4 5 6 "
         "7 8 9 ##0 ##1 ##2 ##9 . , ? ! ' \" ( ) - : ; $ % & / cafe naive resume uber 北 京 東").split()
STRINGS = [
    "Who was the first emperor of Rome?",
    "what year did the war end",
    "  The   capital\tof France\nis Paris!  ",
    "Café naïve résumé ÜBER",
    "don't stop; it's 1999-2000 (approx.) $5.00 & 50%",
    "北京 is the capital; 東京 was new",
    "unplayable replaying songs",
    "xyzzyqq  zero�width​joiner\x00nul\x07bell nbsp",
    "a" * 201 + " short",
    "",
    "[CLS] [SEP] [MASK] brackets",
    "end.",
]


def main():
    vocab_path = os.path.join(HERE, "tokenizer_vocab.txt")
    toks = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"]
    seen = set(toks)
    for w in WORDS:
        if w not in seen:
            toks.append(w); seen.add(w)
    with open(vocab_path, "w", encoding="utf-8") as f:
        f.write("\n".join(toks) + "\n")
    out = {"strings": STRINGS, "cases": {}}
    for extra in (0, 100):
        t = _BertWordPieceTokenizer(vocab_file=vocab_path, lower_case=True, vocab_extra_ids=extra)
        ids = [t.tokenize(s) for s in STRINGS]
        out["cases"][str(extra)] = {
            "ids": ids, "decoded": [t.decode(i) for i in ids], "vocab_size": t.vocab_size, "cls": t.cls, "sep": t.sep, "pad": t.pad,
            "mask": t.mask, "bos": t.bos_token_id, "eos": t.eos_token_id,
            "extra_first": t.vocab.get("<extra_id_0>"), "extra_last": t.vocab.get("<extra_id_99>")}
    tc = _BertWordPieceTokenizer(vocab_file=vocab_path, lower_case=False, vocab_extra_ids=0)
    out["cased_ids"] = [tc.tokenize(s) for s in STRINGS]
    with open(os.path.join(HERE, "tokenizer_ref.json"), "w", encoding="utf-8") as f:
        json.dump(out, f, ensure_ascii=True, indent=0)

    rng = np.random.default_rng(3)
    seqs = [rng.integers(0, 30000, size=int(n)).tolist() for n in (5, 1, 0, 17, 160, 3)]
    prefix = os.path.join(HERE, "mmap_ref")
    b = ref_ds.make_builder(prefix + ".bin", impl="mmap", vocab_size=30522)
    import torch
    for i, s in enumerate(seqs):
        b.add_item(torch.IntTensor(s))
        if i % 2 == 1:
            b.end_document()
    b.finalize(prefix + ".idx")
    d = ref_ds.make_dataset(prefix, impl="mmap", skip_warmup=True)
    assert [x.tolist() for x in d[0:len(seqs)]] == seqs
    with open(prefix + ".json", "w") as f:
        json.dump({"seqs": seqs, "dtype": str(np.dtype(d._index.dtype)), "doc_idx": d.doc_idx.tolist()}, f)
    print("ok", out["cases"]["100"]["vocab_size"], os.path.getsize(prefix + ".bin"), os.path.getsize(prefix + ".idx"))


if __name__ == "__main__":
    main()
