"""Checkpoint plumbing for the entropy models' CDF tables, plus the conv / deconv factories (from ..layers).

`_quantized_cdf`, `_cdf_length` and `_offset` are registered empty and only get their size from `update()`, so a
checkpoint written after `update()` carries shapes the freshly built module does not have yet.  Before
`load_state_dict` the model therefore calls `update_registered_buffers` (same name / arguments as
compressai/models/utils.py:80-110, whose behaviour it reproduces) to give those buffers the checkpoint's shapes.
"""
import torch

from ..layers import conv, deconv  # noqa: F401

_POLICIES = ("resize_if_empty", "resize", "register")


def find_named_buffer(module, query):
    """The buffer registered under the (dotted) name `query`, or None."""
    return dict(module.named_buffers()).get(query)


def update_registered_buffers(module, module_name, buffer_names, state_dict, policy="resize_if_empty", dtype=torch.int):
    """Shape the buffers `buffer_names` of `module` like `state_dict[module_name + "." + name]`.

    policy "resize_if_empty": only buffers that are still empty take the new shape (a table that already exists keeps
    its size and load_state_dict will complain if it disagrees); "resize": always; "register": the buffer must not
    exist yet and is created zero-filled with `dtype`.  Unknown buffer names / policies raise ValueError, a buffer
    that is missing (or, for "register", already there) raises RuntimeError.
    """
    if policy not in _POLICIES:
        raise ValueError(f'Invalid policy "{policy}"')
    have = dict(module.named_buffers())
    unknown = [n for n in buffer_names if n not in have]
    if unknown:
        raise ValueError(f'Invalid buffer name "{unknown[0]}"')
    for name in buffer_names:
        shape = state_dict[f"{module_name}.{name}"].shape
        buf = have.get(name)
        if policy == "register":
            if buf is not None:
                raise RuntimeError(f'buffer "{name}" was already registered')
            module.register_buffer(name, torch.zeros(shape, dtype=dtype))
            continue
        if buf is None:
            raise RuntimeError(f'buffer "{name}" was not registered')
        if policy == "resize" or buf.numel() == 0:
            buf.resize_(shape)
