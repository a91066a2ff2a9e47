"""Task wiring of end-to-end QA training (reference: tasks/openqa/e2eqa/run.py:9-73): dataset, model and metric providers handed to
`train()`.  `main()` is what `tasks/run.py --task OPENQA` dispatches to."""
import csv

import torch

from emdr2_amd import tokenizer as tok
from emdr2_amd.data import indexed_dataset
from emdr2_amd.data.emdr2_index import OpenRetreivalDataStore
from emdr2_amd.data.evidence_arena import EvidenceArena
from emdr2_amd.global_vars import get_args
from emdr2_amd.model.emdr2_model import EMDR2Model, PreComputedEvidenceDocsRetriever
from emdr2_amd.model.transformer import Config
from emdr2_amd.tasks.openqa.e2eqa.train_e2eqa import accuracy_func_provider, print_rank_0, train


def build_tokenizers(args):
    """megatron/global_vars.py:94-110: the retriever's tokenizer and the reader's (same vocabulary + 100 sentinel ids)."""
    from emdr2_amd import global_vars
    if global_vars._ARGS is args:
        return global_vars.get_tokenizer(), global_vars.get_t5_tokenizer()
    bert_tokenizer = tok.build_tokenizer(args, vocab_extra_ids=0)
    args.bert_padded_vocab_size = args.padded_vocab_size
    t5_tokenizer = tok.build_tokenizer(args, vocab_extra_ids=100)
    args.t5_padded_vocab_size = args.padded_vocab_size
    return bert_tokenizer, t5_tokenizer


def read_evidence_titles(evidence_data_path):
    """Title string of every passage in file order (tools/inverted_title_index.py:40-64; file: doc_id <TAB> text <TAB> title, one header)."""
    ids, titles = [], []
    with open(evidence_data_path) as f:
        reader = csv.reader(f, delimiter='\t')
        next(reader, None)
        for row in reader:
            ids.append(int(row[0]))
            titles.append(row[2])
    if ids != list(range(1, len(ids) + 1)):
        raise ValueError("evidence ids must be 1..N in file order (psgs_w100 convention, emdr2_model.py:464-467)")
    return titles


def build_evidence_arena(args):
    passages = indexed_dataset.make_dataset(args.indexed_evidence_data_path, impl=args.data_impl, skip_warmup=not args.mmap_warmup)
    titles = indexed_dataset.make_dataset(args.indexed_title_data_path, impl=args.data_impl, skip_warmup=not args.mmap_warmup)
    return EvidenceArena.from_indexed(passages, titles, read_evidence_titles(args.evidence_data_path)).to_device()


def model_provider(args=None, bert_tokenizer=None, t5_tokenizer=None, arena=None, embed_data=None):
    """run.py:33-39 + emdr2_model.py:31-61,379-408."""
    args = args or get_args()
    print_rank_0('building EMDR2 model for {} ...'.format(args.task))
    if bert_tokenizer is None:
        bert_tokenizer, t5_tokenizer = build_tokenizers(args)
    arena = arena if arena is not None else build_evidence_arena(args)
    retriever = PreComputedEvidenceDocsRetriever(args, arena, embed_data=embed_data)
    cfg = Config(num_layers=args.num_layers, hidden_size=args.hidden_size, num_attention_heads=args.num_attention_heads,
                 ffn_hidden_size=args.ffn_hidden_size, max_position_embeddings=args.max_position_embeddings,
                 layernorm_epsilon=args.layernorm_epsilon, init_method_std=args.init_method_std, hidden_dropout=args.hidden_dropout,
                 attention_dropout=args.attention_dropout, compute_dtype=getattr(args, "compute_dtype", "bf16"))
    torch.manual_seed(args.seed)
    model = EMDR2Model(retriever, cfg, args.t5_padded_vocab_size, args.bert_padded_vocab_size, args.topk_retrievals, args.seq_length,
                       args.seq_length_ret, cls_id=t5_tokenizer.cls, sep_id=t5_tokenizer.sep, pad_id=t5_tokenizer.pad,
                       update_retriever=args.update_retriever, retriever_score_scaling=args.retriever_score_scaling,
                       checkpoint_activations=args.checkpoint_activations and getattr(args, "question_micro_batches", 1) <= 1,
                       disable_retriever_dropout=args.disable_retriever_dropout,
                       no_query_embedder_training=args.no_query_embedder_training,
                       no_context_embedder_training=args.no_context_embedder_training)
    model.set_recompute_keep_last(getattr(args, "recompute_keep_last_layers", 0))
    sel = [int(v) for v in str(getattr(args, "selective_retention_layers", "0,0,0")).split(",")]
    if any(sel):
        model.set_selective_retention(*(sel + [0, 0, 0])[:3])
    return model


def open_retrieval_generative_qa(dataset_cls, arena=None, embed_data=None):
    args = get_args()
    bert_tokenizer, t5_tokenizer = build_tokenizers(args)
    arena = arena if arena is not None else build_evidence_arena(args)

    def train_valid_datasets_provider():
        mk = lambda name, paths: dataset_cls("OPENQA DATASET", name, paths, bert_tokenizer, args.seq_length_ret, args.decoder_seq_length,
                                             seed=args.seed)
        return mk("training", args.train_data), (mk("validation", args.valid_data) if args.valid_data else None)

    def single_dataset_provider(datapath):
        name = datapath[0].split('/')[-1].split('.')[0]
        return dataset_cls("OPENQA_DATASET", name, datapath, bert_tokenizer, args.seq_length_ret, args.decoder_seq_length, seed=args.seed)

    def metrics_provider(datapath):
        return accuracy_func_provider(single_dataset_provider, datapath, t5_tokenizer)

    def indexer_provider(model):
        from emdr2_amd.tasks.openqa.e2eqa.async_indexer import AsyncIndexBuilder
        return AsyncIndexBuilder(model.retriever_model.context_model, arena, model.evidence_retriever.mips_index, args.seq_length_ret,
                                 bert_tokenizer.cls, bert_tokenizer.sep, bert_tokenizer.pad, batch_size=args.indexer_batch_size,
                                 log_interval=args.indexer_log_interval, index_reload_interval=args.index_reload_interval)

    return train(train_valid_datasets_provider, lambda: model_provider(args, bert_tokenizer, t5_tokenizer, arena, embed_data),
                 end_of_epoch_callback_provider=metrics_provider, end_of_training_callback_provider=metrics_provider,
                 eos_id=t5_tokenizer.eos_token_id, indexer_provider=indexer_provider)


def main(arena=None, embed_data=None):
    args = get_args()
    if args.task != "OPENQA":
        raise NotImplementedError('ORQA task {} is not implemented.'.format(args.task))
    from emdr2_amd.tasks.openqa.e2eqa.train_data_utils import OpenQADataset
    return open_retrieval_generative_qa(OpenQADataset, arena=arena, embed_data=embed_data)
