"""Command-line flags of the QA task: the flag set of examples/openqa/emdr2_{nq,trivia,webq}.sh with the reference's names, types and
defaults (megatron/arguments.py:24-596, tasks/run.py:24-67), so those scripts run against this package unchanged.  Flags that configure
machinery this build replaces are accepted and recorded but have no effect: --fp16 (announced at start-up as bf16 activations + fp32
masters, no loss scaling: args.params_dtype), --DDP-impl, --distributed-backend, --num-workers, --faiss-use-gpu (the index always lives in HBM), --max-training-rank /
--async-indexer's extra GPU group (re-indexing runs on a side stream of the trainers), --stale-checkpoint-path, --mmap-warmup.
Unknown flags raise, like argparse in the reference."""
import argparse
import os


def get_parser():
    p = argparse.ArgumentParser(description='EMDR2 on MI355X', allow_abbrev=False)
    g = p.add_argument_group('network size')
    g.add_argument('--num-layers', type=int, default=None)
    g.add_argument('--hidden-size', type=int, default=None)
    g.add_argument('--num-attention-heads', type=int, default=None)
    g.add_argument('--kv-channels', type=int, default=None)
    g.add_argument('--ffn-hidden-size', type=int, default=None)
    g.add_argument('--max-position-embeddings', type=int, default=None)
    g.add_argument('--make-vocab-size-divisible-by', type=int, default=128)
    g.add_argument('--layernorm-epsilon', type=float, default=1e-5)
    g = p.add_argument_group('regularization')
    g.add_argument('--attention-dropout', type=float, default=0.1)
    g.add_argument('--hidden-dropout', type=float, default=0.1)
    g.add_argument('--weight-decay', type=float, default=0.01)
    g.add_argument('--clip-grad', type=float, default=1.0)
    g = p.add_argument_group('training')
    g.add_argument('--batch-size', type=int, default=None)
    g.add_argument('--checkpoint-activations', action='store_true')
    g.add_argument('--recompute-keep-last-layers', type=int, default=0,
                   help='(not in the reference) with --checkpoint-activations: keep the activations of the last N reader-encoder layers '
                        'instead of re-running them in the backward (~33 GB of HBM per layer at batch 64 x top-k 50 x 512 tokens)')
    g.add_argument('--selective-retention-layers', type=str, default='0,0,0',
                   help='(not in the reference) with --checkpoint-activations: READER,CONTEXT,QUERY encoder layers that keep 6 of their ~16 '
                        '[tokens, h] activation tensors and rebuild the rest (both LayerNorm outputs, the FFN pre-activation and GELU output) in '
                        'the backward, instead of being re-run whole (+6.3 GB per reader-encoder layer, +3.0 GB per context-tower layer at '
                        'batch 64 x top-k 50; packed row counts keep growing by 16,384-row steps over the first tens of steps: leave ~25 GB free)')
    g.add_argument('--question-micro-batches', type=int, default=1,
                   help='(not in the reference) run the forward / backward of a step over the batch in M groups of questions '
                        '(EMDR2Model.forward_backward: one MIPS search for the whole batch, everything after it per group, gradients summed in '
                        'the fp32 buckets, losses with batch-wide denominators).  M > 1 keeps every activation of a group instead of re-running '
                        'layers in the backward: --checkpoint-activations is then ignored (batch 64 x top-k 50: M = 4, top-k 100: M = 8)')
    g.add_argument('--seed', type=int, default=1234)
    g.add_argument('--init-method-std', type=float, default=0.02)
    g.add_argument('--lr', type=float, default=None)
    g.add_argument('--lr-decay-style', type=str, default='linear', choices=['linear'])
    g.add_argument('--lr-decay-iters', type=int, default=None)
    g.add_argument('--min-lr', type=float, default=0.0)
    g.add_argument('--warmup', type=float, default=0.01)
    g.add_argument('--log-interval', type=int, default=100)
    g.add_argument('--fp16', action='store_true')
    g.add_argument('--fp32-validation', action='store_true',
                   help='VALIDATION ONLY: run the model in fp32 on the fp32 matrix cores (the reference\'s arithmetic when --fp16 is not given, '
                        'megatron/training.py:55-56): slow, dense layouts, dropout must be 0 -- exists so that "logits within 1e-3 fp32" can be tested')
    g.add_argument('--model-parallel-size', type=int, default=1)
    g.add_argument('--distributed-backend', default='nccl')
    g.add_argument('--DDP-impl', default='local')
    g.add_argument('--local_rank', type=int, default=None)
    g.add_argument('--num-workers', type=int, default=2)
    g = p.add_argument_group('checkpointing')
    g.add_argument('--save', type=str, default=None)
    g.add_argument('--save-interval', type=int, default=None)
    g.add_argument('--load', type=str, default=None)
    g.add_argument('--no-save-optim', action='store_true')
    g.add_argument('--no-load-optim', action='store_true')
    g.add_argument('--pretrained-checkpoint', type=str, default=None)
    g.add_argument('--ict-load', type=str, default=None)
    g.add_argument('--exit-interval', type=int, default=None)
    g.add_argument('--pretrained-t5-load', type=str, default=None)
    g.add_argument('--pretrained-dpr-load', type=str, default=None)
    g.add_argument('--stale-checkpoint-path', type=str, default=None)
    g = p.add_argument_group('validation')
    g.add_argument('--eval-interval', type=int, default=1000)
    g.add_argument('--eval-iters', type=int, default=100)
    g.add_argument('--eval-batch-size', type=int, default=None)
    g.add_argument('--beam-size', type=int, default=1)
    g.add_argument('--max-decode-len', type=int, default=512)
    g = p.add_argument_group('data')
    g.add_argument('--seq-length', type=int, default=None)
    g.add_argument('--seq-length-ret', type=int, default=256)
    g.add_argument('--decoder-seq-length', type=int, default=None)
    g.add_argument('--vocab-file', type=str, default=None)
    g.add_argument('--vocab-extra-ids', type=int, default=0)
    g.add_argument('--tokenizer-type', type=str, default=None, choices=['BertWordPieceLowerCase', 'BertWordPieceCase'])
    g.add_argument('--data-impl', type=str, default='infer', choices=['mmap', 'infer'])
    g.add_argument('--mmap-warmup', action='store_true')
    g.add_argument('--evidence-data-path', type=str, default=None)
    g.add_argument('--indexed-evidence-data-path', type=str, default=None)
    g.add_argument('--indexed-title-data-path', type=str, default=None)
    g.add_argument('--embedding-path', type=str, default=None)
    g.add_argument('--sample-rate', type=float, default=1.0)
    g = p.add_argument_group('task (tasks/run.py)')
    g.add_argument('--task', type=str, required=True)
    g.add_argument('--epochs', type=int, default=None)
    g.add_argument('--keep-last', action='store_true')
    g.add_argument('--train-data', nargs='+', default=None)
    g.add_argument('--valid-data', nargs='*', default=None)
    g.add_argument('--test-data', nargs='*', default=None)
    g = p.add_argument_group('emdr2')
    g.add_argument('--topk-retrievals', type=int, default=100)
    g.add_argument('--emdr2-training', action='store_true')
    g.add_argument('--retriever-score-scaling', action='store_true')
    g.add_argument('--update-retriever', action='store_true')
    g.add_argument('--ret-kldiv', action='store_true')
    g.add_argument('--allow-trivial-doc', action='store_true')
    g.add_argument('--disable-retriever-dropout', action='store_true')
    g.add_argument('--no-query-embedder-training', action='store_true')
    g.add_argument('--no-context-embedder-training', action='store_true')
    g.add_argument('--faiss-use-gpu', action='store_true')
    g.add_argument('--max-training-rank', type=int, default=None)
    g.add_argument('--async-indexer', action='store_true')
    g.add_argument('--index-reload-interval', type=int, default=500)
    g.add_argument('--indexer-batch-size', type=int, default=128)
    g.add_argument('--indexer-log-interval', type=int, default=1000)
    g.add_argument('--report-topk-accuracies', nargs='+', type=int, default=[])
    return p


def parse_args(argv=None):
    args = get_parser().parse_args(argv)
    args.rank = int(os.getenv('RANK', '0'))
    args.world_size = int(os.getenv('WORLD_SIZE', '1'))
    if args.local_rank is None:
        args.local_rank = int(os.getenv('LOCAL_RANK', '0'))
    if args.model_parallel_size != 1:
        raise NotImplementedError("tensor model parallelism is 1 on this path (the reference asserts the same, dualencoder_model.py:15)")
    for name in ('num_layers', 'hidden_size', 'num_attention_heads', 'max_position_embeddings', 'seq_length', 'decoder_seq_length', 'batch_size', 'lr'):
        if getattr(args, name) is None:
            raise ValueError("--%s is required" % name.replace('_', '-'))
    if args.ffn_hidden_size is None:
        args.ffn_hidden_size = 4 * args.hidden_size                       # arguments.py:90-91
    if args.kv_channels is None:
        args.kv_channels = args.hidden_size // args.num_attention_heads
    if args.kv_channels * args.num_attention_heads != args.hidden_size:
        raise ValueError("kv-channels * num-attention-heads must equal hidden-size")
    if args.eval_batch_size is None:
        args.eval_batch_size = args.batch_size
    if args.load and args.ict_load:
        raise ValueError("--load and --ict-load are exclusive (indexer_emdr2.py:47)")
    args.iteration = 0
    # the reference's --fp16 (fp16 activations, fp32 masters inside FP16_Optimizer, dynamic loss scaling) maps to this build's ONLY
    # precision mode: bf16 activations and working weights, fp32 master weights and gradients, no loss scaling.  Said once, recorded in args.
    args.compute_dtype = 'fp32' if getattr(args, 'fp32_validation', False) else 'bf16'
    args.params_dtype = args.compute_dtype
    args.master_dtype = 'fp32'
    if args.rank == 0 and args.compute_dtype == 'fp32':
        print("emdr2_amd: --fp32-validation: fp32 activations and weights on the fp32 matrix cores (validation only: slow, dense layouts, no dropout)",
              flush=True)
    elif args.rank == 0:
        print("emdr2_amd: %sbf16 activations / working weights with fp32 master weights and fp32 weight gradients; no loss scaling "
              "(there is no fp16 compute mode in this build; --fp32-validation runs the slow fp32 validation path)"
              % ("--fp16 requested -> " if args.fp16 else ""), flush=True)
    return args
