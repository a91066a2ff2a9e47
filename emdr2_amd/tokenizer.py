"""BERT WordPiece tokenizer with the reference's extra tokens (megatron/tokenizer/tokenizer.py:133-260, bert_tokenization.py).

The algorithm is the published BERT one: text clean-up (drop control characters, normalise whitespace), spaces around CJK ideographs,
whitespace split, optional lower-casing + accent stripping (NFD, drop Mn), punctuation split, then greedy longest-match-first WordPiece
with the `##` continuation prefix and `[UNK]` for words that cannot be covered or exceed 200 characters.  On top of the vocabulary file
the reference appends `[BOS]`, `[EOS]` and, for the T5-style reader, `vocab_extra_ids` sentinels `<extra_id_i>` -- in that order, which
fixes their ids (bos = len(file), eos = len(file) + 1).  Pinned by tests/golden/tokenizer_ref.json (the reference's tokenizer run on the
same vocabulary and strings).
"""
import collections
import unicodedata


def load_vocab(vocab_file):
    """One token per line, id = line number (bert_tokenization.py:load_vocab)."""
    vocab = collections.OrderedDict()
    with open(vocab_file, "r", encoding="utf-8") as f:
        for index, line in enumerate(f):
            vocab[line.strip()] = index
    return vocab


def _is_whitespace(ch):
    return ch in " \t\n\r" or unicodedata.category(ch) == "Zs"


def _is_control(ch):
    if ch in "\t\n\r":
        return False
    return unicodedata.category(ch) in ("Cc", "Cf")


def _is_punctuation(ch):
    cp = ord(ch)
    if 33 <= cp <= 47 or 58 <= cp <= 64 or 91 <= cp <= 96 or 123 <= cp <= 126:
        return True                                          # ASCII symbols count as punctuation even where Unicode says otherwise ($, ^, `)
    return unicodedata.category(ch).startswith("P")


_CJK_RANGES = ((0x4E00, 0x9FFF), (0x3400, 0x4DBF), (0x20000, 0x2A6DF), (0x2A700, 0x2B73F), (0x2B740, 0x2B81F), (0x2B820, 0x2CEAF),
               (0xF900, 0xFAFF), (0x2F800, 0x2FA1F))


def _is_cjk(cp):
    return any(lo <= cp <= hi for lo, hi in _CJK_RANGES)


class BasicTokenizer(object):
    def __init__(self, do_lower_case=True):
        self.do_lower_case = do_lower_case

    def tokenize(self, text):
        cleaned = []
        for ch in text:
            cp = ord(ch)
            if cp == 0 or cp == 0xFFFD or _is_control(ch):
                continue
            if _is_whitespace(ch):
                cleaned.append(" ")
            elif _is_cjk(cp):
                cleaned.extend((" ", ch, " "))
            else:
                cleaned.append(ch)
        out = []
        for word in "".join(cleaned).split():
            if self.do_lower_case:
                word = "".join(c for c in unicodedata.normalize("NFD", word.lower()) if unicodedata.category(c) != "Mn")
            piece = []
            for ch in word:                                   # every punctuation character is a token of its own
                if _is_punctuation(ch):
                    if piece:
                        out.append("".join(piece)); piece = []
                    out.append(ch)
                else:
                    piece.append(ch)
            if piece:
                out.append("".join(piece))
        return " ".join(out).split()


class WordpieceTokenizer(object):
    def __init__(self, vocab, unk_token="[UNK]", max_input_chars_per_word=200):
        self.vocab, self.unk_token, self.max_chars = vocab, unk_token, max_input_chars_per_word

    def tokenize(self, text):
        out = []
        for word in text.split():
            if len(word) > self.max_chars:
                out.append(self.unk_token)
                continue
            pieces, start, ok = [], 0, True
            while start < len(word):
                end, found = len(word), None
                while start < end:
                    cand = word[start:end] if start == 0 else "##" + word[start:end]
                    if cand in self.vocab:
                        found = cand
                        break
                    end -= 1
                if found is None:
                    ok = False
                    break
                pieces.append(found)
                start = end
            out.extend(pieces if ok else [self.unk_token])
        return out


class FullTokenizer(object):
    def __init__(self, vocab_file, do_lower_case=True):
        self.vocab = load_vocab(vocab_file)
        self.inv_vocab = {v: k for k, v in self.vocab.items()}
        self.basic = BasicTokenizer(do_lower_case)
        self.wordpiece = WordpieceTokenizer(self.vocab)

    def tokenize(self, text):
        out = []
        for tok in self.basic.tokenize(text):
            out.extend(self.wordpiece.tokenize(tok))
        return out

    def convert_tokens_to_ids(self, tokens):
        return [self.vocab[t] for t in tokens]

    def convert_ids_to_tokens(self, ids):
        # ids in the padded tail of the embedding table (vocab_size_with_padding) have no token: an untrained reader can emit them
        return [self.inv_vocab.get(int(i), "[UNK]") for i in ids]

    @staticmethod
    def convert_tokens_to_string(tokens, clean_up_tokenization_spaces=True):
        text = " ".join(tokens).replace(" ##", "").strip()
        if clean_up_tokenization_spaces:
            for a, b in ((" .", "."), (" ?", "?"), (" !", "!"), (" ,", ","), (" ' ", "'"), (" n't", "n't"), (" 'm", "'m"), (" 's", "'s"),
                         (" 've", "'ve"), (" 're", "'re")):
                text = text.replace(a, b)
        return text

    def vocab_size(self):
        return len(self.vocab)


class BertWordPieceTokenizer(object):
    """`_BertWordPieceTokenizer` (tokenizer.py:133-260): `.tokenize(text) -> ids`, `.decode(ids) -> str`, cls / sep / pad / mask,
    bos_token_id / eos_token_id, `vocab_extra_ids` sentinels for the reader's vocabulary."""

    def __init__(self, vocab_file, lower_case=True, vocab_extra_ids=0):
        self.tokenizer = FullTokenizer(vocab_file, do_lower_case=lower_case)
        v = self.tokenizer.vocab
        self.cls_id, self.sep_id, self.pad_id, self.mask_id = v["[CLS]"], v["[SEP]"], v["[PAD]"], v["[MASK]"]
        self._bos_token, self._eos_token = "[BOS]", "[EOS]"
        self.add_token(self._bos_token)
        self.add_token(self._eos_token)
        self._bos_token_id, self._eos_token_id = v[self._bos_token], v[self._eos_token]
        self.additional_special_tokens = ["<extra_id_{}>".format(i) for i in range(vocab_extra_ids)]
        for t in self.additional_special_tokens:
            self.add_token(t)

    def add_token(self, token):
        if token not in self.vocab:
            self.inv_vocab[self.vocab_size] = token
            self.vocab[token] = self.vocab_size

    @property
    def vocab_size(self):
        return self.tokenizer.vocab_size()

    @property
    def vocab(self):
        return self.tokenizer.vocab

    @property
    def inv_vocab(self):
        return self.tokenizer.inv_vocab

    def tokenize(self, text):
        return self.tokenizer.convert_tokens_to_ids(self.tokenizer.tokenize(text))

    def decode(self, ids):
        return self.tokenizer.convert_tokens_to_string(self.tokenizer.convert_ids_to_tokens(ids))

    cls = property(lambda self: self.cls_id)
    sep = property(lambda self: self.sep_id)
    pad = property(lambda self: self.pad_id)
    mask = property(lambda self: self.mask_id)
    bos_token_id = property(lambda self: self._bos_token_id)
    eos_token_id = property(lambda self: self._eos_token_id)


def vocab_size_with_padding(orig_vocab_size, make_vocab_size_divisible_by=128, model_parallel_size=1):
    """tokenizer.py:57-70."""
    multiple = make_vocab_size_divisible_by * model_parallel_size
    return ((orig_vocab_size + multiple - 1) // multiple) * multiple


def build_tokenizer(args, vocab_extra_ids=0):
    """tokenizer.py:23-54: sets args.padded_vocab_size."""
    if getattr(args, "vocab_extra_ids", 0) > 0:
        vocab_extra_ids = args.vocab_extra_ids
    if args.tokenizer_type not in ("BertWordPieceLowerCase", "BertWordPieceCase"):
        raise NotImplementedError("{} tokenizer is not implemented.".format(args.tokenizer_type))
    tok = BertWordPieceTokenizer(args.vocab_file, lower_case=args.tokenizer_type == "BertWordPieceLowerCase", vocab_extra_ids=vocab_extra_ids)
    args.padded_vocab_size = vocab_size_with_padding(tok.vocab_size, getattr(args, "make_vocab_size_divisible_by", 128),
                                                     getattr(args, "model_parallel_size", 1))
    return tok
