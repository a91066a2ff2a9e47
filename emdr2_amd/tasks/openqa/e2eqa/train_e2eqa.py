"""Training-step driver of the QA task (reference: tasks/openqa/e2eqa/train_e2eqa.py:28-41,126-181 and
megatron/training.py:165-230).  Same forward-step contract: forward_step(batch_or_iter, model) -> (loss, {'lm_loss', 'retriever_loss'})."""
import torch

from emdr2_amd.model.emdr2_model import emdr2_loss
from emdr2_amd.training import allreduce_gradients


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
    lm_logits, topk_log_probs, lm_logits_one_context = model(query_uid, q_bert, q_types, q_mask, q_t5, q_t5_len, dec_ids)
    net_loss, stats = emdr2_loss(lm_logits, topk_log_probs, lm_logits_one_context, labels, loss_mask, eos_id)
    return net_loss, {'lm_loss': stats['lm_loss'], 'retriever_loss': stats['retriever_loss']}


def train_step(forward_step_func, data_iterator, model, optimizer, lr_scheduler, eos_id, dp_group=None):
    """megatron/training.py:202-230 without the fp16 machinery (bf16 needs no loss scale / overflow skip)."""
    optimizer.zero_grad()
    loss, loss_reduced = forward_step_func(data_iterator, model, eos_id)
    loss.backward()
    allreduce_gradients(model, dp_group)
    optimizer.step(lr=lr_scheduler.step())
    return loss_reduced
