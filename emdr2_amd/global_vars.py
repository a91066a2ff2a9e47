"""Process-wide argument singleton (reference: megatron/global_vars.py:35-38 `get_args`).  Only the
fields the hot path reads are needed; `set_args` accepts any namespace-like object (e.g. the
reference's parsed args) so the modules drop in under the reference's entry points."""
_ARGS = None


def set_args(args):
    global _ARGS
    _ARGS = args


def get_args():
    if _ARGS is None:
        raise RuntimeError("emdr2_amd.global_vars: args are not initialised (call set_args)")
    return _ARGS
