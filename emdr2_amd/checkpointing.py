"""Checkpoint I/O in the reference's on-disk layout (megatron/checkpointing.py:74-135,267-340), so released EMDR2 / pre-trained T5 and
dual-encoder checkpoints load directly and checkpoints written here load in the reference:

    <dir>/latest_checkpointed_iteration.txt                      the tracker: an iteration number or 'release'
    <dir>/iter_{:07d}/mp_rank_00/model_optim_rng.pt              torch.save({'iteration', 'model', 'optimizer', 'lr_scheduler', ...})

'model' is the NESTED dict the reference's `state_dict_for_save_checkpoint()` methods build (emdr2_model.py:217-231, t5_model.py:156-168,
dualencoder_model.py:84-98,183-189, language_model.py:183-198,367-387):

    EMDR2   {'encoder/t5_model': T5, 'retriever/biencoder_model': {'query_model': BERT, 'context_model': BERT}}
    T5      {'language_model': LM, 'lm_head': {'bias'}}             BERT {'language_model': LM}
    LM      {'embedding': {'word_embeddings': {'weight'}, 'position_embeddings': {'weight'}, 'tokentype_embeddings': {'weight'}},
             'encoder': {flat ParallelTransformer keys}, 'decoder': {...}}

Key paths and shapes are pinned against the reference's own model in tests/golden/ckpt_layout.json.  Tensors are stored as fp32 masters
(the reference stores fp16 weights + fp32 masters inside the optimizer; both load here: values are cast to fp32)."""
import os

import torch


def get_checkpoint_name(checkpoints_path, iteration, release=False, mp_rank=0):
    directory = 'release' if release else 'iter_{:07d}'.format(iteration)
    return os.path.join(checkpoints_path, directory, 'mp_rank_{:02d}'.format(mp_rank), 'model_optim_rng.pt')


def get_checkpoint_tracker_filename(checkpoints_path):
    return os.path.join(checkpoints_path, 'latest_checkpointed_iteration.txt')


# ---- nested <-> module --------------------------------------------------------------------------------------------------------
def _lm_nested(lm):
    emb = lm.embedding
    e = {'word_embeddings': emb.word_embeddings.state_dict(), 'position_embeddings': emb.position_embeddings.state_dict()}
    if emb.tokentype_embeddings is not None:
        e['tokentype_embeddings'] = emb.tokentype_embeddings.state_dict()
    out = {'embedding': e, 'encoder': lm.encoder.state_dict()}
    if lm.add_decoder:
        out['decoder'] = lm.decoder.state_dict()
    return out


def _load(module, sd, strict=True):
    sd = {k: v.to(torch.float32) for k, v in sd.items()}
    module.load_state_dict(sd, strict=strict)


def _lm_load(lm, nested, strict=True):
    e = nested['embedding']
    _load(lm.embedding.word_embeddings, e['word_embeddings'], strict)
    _load(lm.embedding.position_embeddings, e['position_embeddings'], strict)
    if lm.embedding.tokentype_embeddings is not None and 'tokentype_embeddings' in e:
        _load(lm.embedding.tokentype_embeddings, e['tokentype_embeddings'], strict)
    _load(lm.encoder, nested['encoder'] if 'encoder' in nested else nested['transformer'], strict)
    if lm.add_decoder:
        _load(lm.decoder, nested['decoder'], strict)


def t5_state_dict(t5):
    return {'language_model': _lm_nested(t5.language_model), 'lm_head': t5.lm_head.state_dict()}


def load_t5_state_dict(t5, nested, strict=True):
    _lm_load(t5.language_model, nested['language_model'], strict)
    _load(t5.lm_head, nested['lm_head'], strict)


def dualencoder_state_dict(de):
    return {'query_model': {'language_model': _lm_nested(de.query_model.language_model)},
            'context_model': {'language_model': _lm_nested(de.context_model.language_model)}}


def load_dualencoder_state_dict(de, nested, strict=True, only_query_model=False, only_context_model=False):
    if not only_context_model:
        _lm_load(de.query_model.language_model, nested['query_model']['language_model'], strict)
    if not only_query_model:
        _lm_load(de.context_model.language_model, nested['context_model']['language_model'], strict)


def emdr2_state_dict(model):
    return {model._language_model_key: t5_state_dict(model.language_model), model._retriever_model_key: dualencoder_state_dict(model.retriever_model)}


def load_emdr2_state_dict(model, nested, strict=True):
    load_t5_state_dict(model.language_model, nested[model._language_model_key], strict)
    load_dualencoder_state_dict(model.retriever_model, nested[model._retriever_model_key], strict)


# ---- files --------------------------------------------------------------------------------------------------------------------
def _invalidate_weight_caches():
    from emdr2_amd.model import kernels
    kernels.WEIGHTS.invalidate()


def save_checkpoint(save_dir, iteration, model, optimizer=None, lr_scheduler=None, rank=0, barrier=None, args=None):
    """checkpointing.py:94-135: data-parallel rank 0 writes, then the tracker is updated."""
    if rank == 0:
        name = get_checkpoint_name(save_dir, iteration)
        os.makedirs(os.path.dirname(name), exist_ok=True)
        # checkpoint_version 1.0 = the QKV / KV projections are stored in the [np, hn, 3] / [np, hn, 2] interleaved row order
        # (transformer.py:225-259); without the key the reference assumes version 0 and re-orders the rows on load (checkpointing.py:236).
        state = {'checkpoint_version': 1.0, 'iteration': iteration, 'model': emdr2_state_dict(model)}
        if args is not None:
            state['args'] = args
        if optimizer is not None:
            state['optimizer'] = optimizer.state_dict()
        if lr_scheduler is not None:
            state['lr_scheduler'] = lr_scheduler.state_dict()
        torch.save(state, name)
    if barrier is not None:
        barrier()
    if rank == 0:
        with open(get_checkpoint_tracker_filename(save_dir), 'w') as f:
            f.write(str(iteration))
    if barrier is not None:
        barrier()


def read_tracker(load_dir):
    """(iteration, release) or (0, False) when there is no checkpoint (checkpointing.py:150-175)."""
    tracker = get_checkpoint_tracker_filename(load_dir) if load_dir else None
    if not tracker or not os.path.isfile(tracker):
        return 0, False
    s = open(tracker).read().strip()
    if s == 'release':
        return 0, True
    return int(s), False


def _read(load_dir, want_release=False):
    iteration, release = read_tracker(load_dir)
    if iteration == 0 and not release:
        return (None, 0, False) if want_release else (None, 0)
    state = torch.load(get_checkpoint_name(load_dir, iteration, release), map_location='cpu', weights_only=False)
    return (state, iteration, release) if want_release else (state, iteration)


def _check_version(state):
    """The modules here read the version-1.0 row order only.  Version 0 files (no 'checkpoint_version' key written by an old Megatron:
    rows grouped [3, np, hn]) would load silently with scrambled Q/K/V rows; the reference converts them on the fly
    (transformer.py:225-248), this loader refuses them."""
    version = state.get('checkpoint_version', 0)
    if version < 1.0:
        raise ValueError("checkpoint_version %s: the QKV rows of this file are not in the [np, hn, 3] order this loader reads; "
                         "re-save it with the reference (which converts version 0 on load) first" % version)


def load_checkpoint(load_dir, model, optimizer=None, lr_scheduler=None):
    """Resume (checkpointing.py:138-263); returns the iteration (0 = nothing to load)."""
    state, iteration, release = _read(load_dir, want_release=True)
    if state is None:
        return 0
    # only this path consults the version, like the reference (checkpointing.py:202 -> set_checkpoint_version): its pretrained-model loaders
    # (:267-340) never do, so a pretrained T5 / DPR file without the key is read in the version-1.0 row order there and here
    _check_version(state)
    load_emdr2_state_dict(model, state['model'])
    _invalidate_weight_caches()
    if release:
        # a released checkpoint starts a NEW run: iteration 0, fresh optimizer and schedule (checkpointing.py:196-204,243-256)
        return 0
    if optimizer is not None and 'optimizer' in state:
        try:
            optimizer.load_state_dict(state['optimizer'])
        except (KeyError, TypeError, AttributeError) as exc:
            raise ValueError("the optimizer state of this checkpoint is not in this package's format (a reference FP16_Optimizer state?): "
                             "load the weights only with --no-load-optim (%r)" % (exc,))
        if lr_scheduler is not None and 'lr_scheduler' in state:      # the reference loads the schedule only together with the optimizer
            lr_scheduler.load_state_dict(state['lr_scheduler'])
    return state.get('iteration', iteration)


def load_t5_checkpoint(t5, custom_load_path):
    """checkpointing.py:308-340."""
    state, _ = _read(custom_load_path)
    if state is None:
        raise FileNotFoundError("no checkpoint under %s" % custom_load_path)
    load_t5_state_dict(t5, state['model'])
    _invalidate_weight_caches()


def load_dualencoder_checkpoint(de, custom_load_path, key_list=None, only_query_model=False, only_context_model=False):
    """checkpointing.py:267-305: `key_list` walks into a combined checkpoint (['retriever/biencoder_model'] for an EMDR2 one)."""
    state, _ = _read(custom_load_path)
    if state is None:
        raise FileNotFoundError("no checkpoint under %s" % custom_load_path)
    sd = state['model']
    for key in key_list or []:
        sd = sd[key]
    load_dualencoder_state_dict(de, sd, only_query_model=only_query_model, only_context_model=only_context_model)
    _invalidate_weight_caches()
