"""Process-wide singletons (reference: megatron/global_vars.py:35-110 `get_args`, `get_tokenizer`, `get_t5_tokenizer`).  Only the
fields the hot path reads are needed; `set_args` accepts any namespace-like object (e.g. the reference's parsed args) so the modules
drop in under the reference's entry points."""
_ARGS = None
_TOKENIZERS = None


def set_args(args):
    global _ARGS, _TOKENIZERS
    _ARGS, _TOKENIZERS = args, None


def get_args():
    if _ARGS is None:
        raise RuntimeError("emdr2_amd.global_vars: args are not initialised (call set_args)")
    return _ARGS


def _tokenizers():
    """(retriever tokenizer, reader tokenizer = same vocabulary + 100 sentinel ids), built once from args.vocab_file; also records
    args.bert_padded_vocab_size / args.t5_padded_vocab_size like the reference's set_global_variables (global_vars.py:94-110)."""
    global _TOKENIZERS
    if _TOKENIZERS is None:
        from emdr2_amd import tokenizer as tok
        args = get_args()
        bert = tok.build_tokenizer(args, vocab_extra_ids=0)
        args.bert_padded_vocab_size = args.padded_vocab_size
        t5 = tok.build_tokenizer(args, vocab_extra_ids=100)
        args.t5_padded_vocab_size = args.padded_vocab_size
        _TOKENIZERS = (bert, t5)
    return _TOKENIZERS


def get_tokenizer():
    return _tokenizers()[0]


def get_t5_tokenizer():
    return _tokenizers()[1]
