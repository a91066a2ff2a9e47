"""Training-step driver of the QA task (reference: tasks/openqa/e2eqa/train_e2eqa.py:28-41,126-181 and
megatron/training.py:165-230).  Same forward-step contract: forward_step(batch_or_iter, model) -> (loss, {'lm_loss', 'retriever_loss'})."""
import time

import torch

from emdr2_amd import checkpointing
from emdr2_amd.global_vars import get_args
from emdr2_amd.model.emdr2_model import emdr2_loss
from emdr2_amd.tasks.openqa.e2eqa.eval_utils import exact_match_score, metric_max_over_ground_truths
from emdr2_amd.tasks.openqa.e2eqa.train_data_utils import collate
from emdr2_amd.training import AnnealingLR, allreduce_gradients


def process_batch(batch):
    dev = "cuda"
    return (batch['query_uid'].to(dev), batch['query_ids_bert'].to(dev), batch['query_types'].to(dev), None,
            batch['query_ids_t5'].to(dev), batch['query_ids_t5_len'].to(dev), batch['dec_ids'].to(dev), batch['labels'].to(dev),
            batch['loss_mask'].to(dev), batch.get('reference'))


def _cross_entropy_forward_step(batch, model, eos_id):
    try:
        batch_ = next(batch)
    except BaseException:
        batch_ = batch
    query_uid, q_bert, q_types, q_mask, q_t5, q_t5_len, dec_ids, labels, loss_mask, _ = process_batch(batch_)
    assert torch.all(query_uid < 0), "query uid can't be positive"
    try:
        ret_kldiv, micro = bool(getattr(get_args(), 'ret_kldiv', False)), int(getattr(get_args(), 'question_micro_batches', 1) or 1)
    except RuntimeError:
        ret_kldiv, micro = False, 1
    if micro > 1 and model.training:
        # --question-micro-batches: forward, loss and backward per group of questions inside the model; the returned loss carries no graph
        # (train_step skips its backward call)
        net_loss, stats = model.forward_backward(query_uid, q_bert, q_types, q_mask, q_t5, q_t5_len, dec_ids, labels, loss_mask, eos_id,
                                                 micro_batches=micro, ret_kldiv=ret_kldiv)
        return net_loss, {'lm_loss': stats['lm_loss'], 'retriever_loss': stats['retriever_loss']}
    lm_logits, topk_log_probs, lm_logits_one_context = model(query_uid, q_bert, q_types, q_mask, q_t5, q_t5_len, dec_ids)
    net_loss, stats = emdr2_loss(lm_logits, topk_log_probs, lm_logits_one_context, labels, loss_mask, eos_id, ret_kldiv=ret_kldiv)
    return net_loss, {'lm_loss': stats['lm_loss'], 'retriever_loss': stats['retriever_loss']}


def train_step(forward_step_func, data_iterator, model, optimizer, lr_scheduler, eos_id, dp_group=None, guard=None):
    """megatron/training.py:202-230 without the fp16 machinery (bf16 needs no loss scale / overflow skip).  `guard` (training.RetentionGuard,
    built when --recompute-keep-last-layers / --selective-retention-layers ask for activations to be kept or --question-micro-batches
    splits the step): a step that runs out of HBM is run again on all ranks together, with a thinner retention plan / a finer split if
    need be, instead of ending the job."""
    from emdr2_amd.model import kernels
    try:
        batch = next(data_iterator)                  # fetched once: a re-run of the step sees the same batch
    except TypeError:
        batch = data_iterator

    def step_once():
        optimizer.zero_grad()
        sink = kernels.GRAD_SINK
        if sink is not None and sink is not optimizer:
            sink.begin_step()
        loss, loss_reduced = forward_step_func(batch, model, eos_id)
        if loss.requires_grad:                       # (a micro-batched forward step has already back-propagated, group by group)
            loss.backward()
        if sink is not None:
            sink.finish()                            # buckets were all-reduced (bf16, pre-divided) while the backward ran
        else:
            allreduce_gradients(model, dp_group)
        return loss_reduced
    loss_reduced = guard.run(step_once) if guard is not None else step_once()
    # the reference steps the optimizer with the rate its scheduler set at the END of the previous iteration (training.py:223-228,
    # learning_rates.py:73-80): step n runs at lr(n - 1), so the very first update of a warm-up schedule has lr 0
    optimizer.step(lr=lr_scheduler.get_lr())
    lr_scheduler.step()
    return loss_reduced


# =====================================================================================================================================
# the task driver: data loaders, evaluation callbacks, epoch loop (reference: tasks/openqa/e2eqa/train_e2eqa.py:216-621)
# =====================================================================================================================================
def print_rank_0(msg):
    if not torch.distributed.is_initialized() or torch.distributed.get_rank() == 0:
        print(msg, flush=True)


def _dp():
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        return torch.distributed.get_rank(), torch.distributed.get_world_size()
    return 0, 1


def build_data_loader(dataset, batch_size, num_workers, drop_last, shuffle=True):
    """train_e2eqa.py:343-367: per-rank batches through a DistributedSampler; the collate keeps the 10-key batch of a1."""
    rank, world = _dp()
    sampler = torch.utils.data.distributed.DistributedSampler(dataset, num_replicas=world, rank=rank, shuffle=shuffle)
    return torch.utils.data.DataLoader(dataset, batch_size=batch_size, sampler=sampler, shuffle=False, num_workers=num_workers,
                                       drop_last=drop_last, pin_memory=True, collate_fn=collate)


def reader_em_score(model, dataloader, topk_retrievals, tokenizer):
    """train_e2eqa.py:216-283: greedy (--beam-size 1) or beam-search answers, exact match against every reference answer, de-duplicated
    over ranks by query uid."""
    from emdr2_amd.model.search_strategy import BeamSearch, SampleOrGreedySearch
    args = get_args()
    if args.beam_size < 1:
        raise AssertionError("--beam-size < 1 is not supported for ORQA reader.")
    score_list, quid_list = [], []
    model.eval()
    with torch.no_grad():
        for batch in dataloader:
            query_uid, q_bert, q_types, q_mask, q_t5, q_t5_len, _, _, _, reference = process_batch(batch)
            if args.beam_size == 1:
                search = SampleOrGreedySearch(max_decode_len=args.max_decode_len, bos_id=tokenizer.bos_token_id, eos_id=tokenizer.eos_token_id,
                                              sample=False, topk_evidence=topk_retrievals)
            else:
                search = BeamSearch(max_decode_len=args.max_decode_len, bos_id=tokenizer.bos_token_id, eos_id=tokenizer.eos_token_id,
                                    beam_size=args.beam_size, topk_evidence=topk_retrievals)
            hypothesis = search.generate_output(model, query_uid, q_bert, q_types, q_mask, q_t5, q_t5_len)
            for quid, ref, hyp in zip(query_uid.tolist(), reference, hypothesis):
                score_list.append(float(metric_max_over_ground_truths(exact_match_score, tokenizer.decode(hyp), ref)))
                quid_list.append(quid)
    model.train()
    scores = torch.tensor(score_list, dtype=torch.float32, device="cuda")
    quids = torch.tensor(quid_list, dtype=torch.int64, device="cuda")
    rank, world = _dp()
    if world > 1:                                                        # ranks may hold different counts (drop_last False): pad to the max
        n = torch.tensor([scores.numel()], device="cuda"); nmax = n.clone()
        torch.distributed.all_reduce(nmax, op=torch.distributed.ReduceOp.MAX)
        pad = int(nmax.item()) - scores.numel()
        scores = torch.cat([scores, torch.zeros(pad, device="cuda")]); quids = torch.cat([quids, torch.zeros(pad, dtype=torch.int64, device="cuda")])
        gs = [torch.empty_like(scores) for _ in range(world)]; gq = [torch.empty_like(quids) for _ in range(world)]
        torch.distributed.all_gather(gs, scores); torch.distributed.all_gather(gq, quids)
        scores, quids = torch.cat(gs), torch.cat(gq)
    score_dict = {q: s for q, s in zip(quids.tolist(), scores.tolist()) if q != 0}     # duplicates (sampler padding) overwrite
    return {'Exact Match Score': sum(score_dict.values())}, len(score_dict)


def validation_loss(model, dataloader, eos_id):
    """train_e2eqa.py:286-321."""
    total, score = 0, 0.0
    model.eval()
    with torch.no_grad():
        for batch in dataloader:
            query_uid, q_bert, q_types, q_mask, q_t5, q_t5_len, dec_ids, labels, loss_mask, _ = process_batch(batch)
            lm_logits, topk_log_probs, _, _ = model(query_uid, q_bert, q_types, q_mask, q_t5, q_t5_len, dec_ids)
            _, stats = emdr2_loss(lm_logits, topk_log_probs, None, labels, loss_mask, eos_id)
            total += 1
            score += float(stats['lm_loss'])
    model.train()
    t = torch.tensor([score, total], dtype=torch.float32, device="cuda")
    if _dp()[1] > 1:
        torch.distributed.all_reduce(t)
    return {'Validation Loss': t[0]}, t[1]


def accuracy_func_provider(single_dataset_provider, datapath, tokenizer):
    """train_e2eqa.py:324-341."""
    args = get_args()
    if not datapath:
        return lambda model, epoch: None
    dataset = single_dataset_provider(datapath)
    dataloader = build_data_loader(dataset, args.eval_batch_size, num_workers=args.num_workers, drop_last=False, shuffle=False)

    def metrics_func(model, epoch):
        print_rank_0('calculating metrics ...')
        stats, total = reader_em_score(model, dataloader, args.topk_retrievals, tokenizer)
        fmt = "|total_questions: {}".format(total)
        for k, v in stats.items():
            fmt += "|{} = {:.2f}".format(k, (v * 100) / max(total, 1))
        print_rank_0("epoch:{}{}".format(epoch, fmt))
        return stats, total
    return metrics_func


def setup_model_and_optimizer(model_provider):
    from emdr2_amd.model import kernels as _K
    _K.PACKING.sticky = True                                  # training: packed stacks keep constant sizes from step to step (kernels._Packing)
    return _setup_model_and_optimizer(model_provider)


def _setup_model_and_optimizer(model_provider):
    """megatron/training.py:136-162: model, FusedAdam over the (decay / no-decay) groups, AnnealingLR, resume or pre-trained init."""
    args = get_args()
    model = model_provider()
    from emdr2_amd.model import kernels
    from emdr2_amd.training import FlatAdam
    # flat buckets (masters, gradients, moments, bf16 working copies), no-decay grouping of model/utils.py:64-83 inside each bucket; it is
    # also the gradient sink of the backward: data-parallel buckets are exchanged in bf16 as soon as they are complete
    optimizer = kernels.GRAD_SINK = FlatAdam(model, lr=args.lr, weight_decay=args.weight_decay, clip_grad=args.clip_grad)
    num_iters = args.lr_decay_iters if args.lr_decay_iters is not None else args.train_iters
    num_iters = max(1, num_iters)
    lr_scheduler = AnnealingLR(args.lr, warmup_iter=args.warmup * num_iters, total_iters=num_iters, min_lr=args.min_lr)
    no_optim = args.no_load_optim or getattr(args, 'finetune', False)                  # checkpointing.py:243-256: schedule only with the optimizer
    args.iteration = checkpointing.load_checkpoint(args.load, model, None if no_optim else optimizer, None if no_optim else lr_scheduler) if args.load else 0
    if args.iteration == 0:
        model.init_state_dict_from_dpr_and_t5(args.pretrained_t5_load, args.pretrained_dpr_load)      # training.py:156-158
    return model, optimizer, lr_scheduler


def _save(iteration, model, optimizer, lr_scheduler):
    args = get_args()
    rank, world = _dp()
    checkpointing.save_checkpoint(args.save, iteration, model, None if args.no_save_optim else optimizer, lr_scheduler, rank=rank,
                                  barrier=torch.distributed.barrier if world > 1 else None, args=args)


def _train(model, optimizer, lr_scheduler, forward_step, train_dataloader, end_of_epoch_callback, end_of_epoch_callback2, eos_id, indexer=None):
    """train_e2eqa.py:413-552.  `indexer`: an AsyncIndexBuilder when --async-indexer (re-indexing on a side stream; the swap at a step
    boundary replaces the NEW_INDEX_READY / NEW_CHKPT_READY handshake and the reload from disk)."""
    args = get_args()
    model.train()
    guard = None
    sel = [int(v) for v in str(getattr(args, "selective_retention_layers", "0,0,0")).split(",")] + [0, 0, 0]
    keep = int(getattr(args, "recompute_keep_last_layers", 0) or 0)
    micro = int(getattr(args, "question_micro_batches", 1) or 1)
    if keep or any(sel[:3]) or micro > 1:
        from emdr2_amd.training import RetentionGuard
        retr = getattr(model, "evidence_retriever", None)

        def set_micro(m):                             # _cross_entropy_forward_step reads the flag: a finer split after an allocation failure
            args.question_micro_batches = m           # takes effect in the re-run of the very step that failed
        guard = RetentionGuard(model, optimizer, keep=keep, reader=sel[0], context=sel[1], query=sel[2],
                               forward_progress=(lambda: getattr(retr, "searches", 0)) if retr is not None else None, log=print_rank_0,
                               micro=micro, batch=args.batch_size, on_micro_change=set_micro)
    start_epoch = args.iteration // args.train_iters_per_epoch
    start_iteration = args.iteration % args.train_iters_per_epoch
    iteration = args.iteration
    sums, t_last = {}, time.time()
    for epoch in range(start_epoch, args.epochs):
        print_rank_0('working on epoch {} ...'.format(epoch + 1))
        train_dataloader.sampler.set_epoch(args.seed + epoch)
        for iteration_, batch in enumerate(train_dataloader):
            if iteration_ < start_iteration:
                continue
            start_iteration = 0
            if indexer is not None:
                indexer.pump()
            losses = train_step(forward_step, batch, model, optimizer, lr_scheduler, eos_id, guard=guard)
            iteration += 1
            if indexer is not None and indexer.maybe_swap(iteration):
                print_rank_0("Training Group: MIPS Index Updated at iteration {}".format(iteration))
                if args.save:
                    _save(iteration, model, optimizer, lr_scheduler)                          # the reference checkpoints at every reload
            for k, v in losses.items():
                sums[k] = sums.get(k, 0.0) + v
            if iteration % args.log_interval == 0:
                dt = (time.time() - t_last) * 1000.0 / args.log_interval
                t_last = time.time()
                msg = ' iteration {:8d}/{:8d} | elapsed time per iteration (ms): {:.1f} | learning rate: {:.3E} |'.format(
                    iteration, args.train_iters, dt, lr_scheduler.get_lr())
                for k in sums:
                    msg += ' {}: {:.6E} |'.format(k, float(sums[k]) / args.log_interval)
                print_rank_0(msg)
                sums = {}
            if args.save and args.save_interval and iteration % args.save_interval == 0:
                _save(iteration, model, optimizer, lr_scheduler)
            if args.eval_interval and iteration % args.eval_interval == 0 and end_of_epoch_callback is not None:
                end_of_epoch_callback(model, iteration)
                end_of_epoch_callback2(model, iteration)
            if args.exit_interval and iteration % args.exit_interval == 0:                    # train_e2eqa.py:526-540
                print_rank_0('exiting the program at iteration {}'.format(iteration))
                if args.save:
                    _save(iteration, model, optimizer, lr_scheduler)
                return iteration
        if args.save:
            _save(iteration, model, optimizer, lr_scheduler)
        if end_of_epoch_callback is not None:
            end_of_epoch_callback(model, epoch + 1)
            end_of_epoch_callback2(model, epoch + 1)
    return iteration


def train(train_valid_datasets_provider, model_provider, forward_step=_cross_entropy_forward_step, end_of_epoch_callback_provider=None,
          end_of_training_callback_provider=None, eos_id=None, indexer_provider=None):
    """train_e2eqa.py:555-621."""
    args = get_args()
    train_dataloader = None
    if args.epochs > 0:
        train_dataset, valid_dataset = train_valid_datasets_provider()
        train_dataloader = build_data_loader(train_dataset, args.batch_size, args.num_workers, drop_last=not args.keep_last)
        args.train_iters_per_epoch = len(train_dataloader)
        args.train_iters = args.epochs * args.train_iters_per_epoch
    else:
        args.train_iters_per_epoch, args.train_iters = 1, 0
    cb1 = cb2 = None
    if args.epochs > 0 and end_of_epoch_callback_provider is not None:
        cb1, cb2 = end_of_epoch_callback_provider(args.valid_data), end_of_epoch_callback_provider(args.test_data)
    model, optimizer, lr_scheduler = setup_model_and_optimizer(model_provider)
    if args.iteration == 0 and args.pretrained_checkpoint is not None:                     # train_e2eqa.py:586-593: weights only
        checkpointing.load_checkpoint(args.pretrained_checkpoint, model, None, None)
    print_rank_0('done with setups ...')
    print_rank_0('training ...')
    if args.epochs > 0 and args.emdr2_training:
        indexer = indexer_provider(model) if (indexer_provider is not None and args.async_indexer) else None
        _train(model, optimizer, lr_scheduler, forward_step, train_dataloader, cb1, cb2, eos_id, indexer)
    results = {}
    if end_of_training_callback_provider is not None:
        for name, path in (("validation", args.valid_data), ("test", args.test_data)):
            if path:
                results[name] = end_of_training_callback_provider(path)(model, epoch=-1)
    print_rank_0('done :-)')
    return model, results
