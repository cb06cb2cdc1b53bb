"""Forward-hook instrumentation of a torch module tree.

Host-side counterpart of the reference's util/nethook.py (``Trace`` :22-128, ``TraceDict`` :131-200,
``get_module`` :375, ``get_parameter`` :385, ``set_requires_grad`` :360) built on plain
``register_forward_hook``.  Dotted names resolve with or without the ``text_model.`` prefix, because
transformers 4.x (the reference's pin) nests the CLIP text tower under ``text_model`` and 5.x does not.
"""
import contextlib
from collections import OrderedDict

import torch


class StopForward(Exception):
    """Raised by a hook to abandon the rest of the forward pass (reference: nethook.py ``stop=True``)."""


def _candidates(name):
    yield name
    if name.startswith("text_model."):
        yield name[len("text_model."):]
    else:
        yield "text_model." + name


def _walk(model, name):
    """The sub-module at dotted path ``name`` by walking ``_modules`` (O(depth)); None when the path does not exist or
    the object is not built from ``torch.nn.Module``s (duck-typed encoders take the ``named_*`` tables below)."""
    obj = model
    if name:
        for part in name.split("."):
            mods = getattr(obj, "_modules", None)
            if mods is None:
                return None
            obj = mods.get(part)
            if obj is None:
                return None
    return obj


def get_module(model, name):
    """``dict(model.named_modules())[name]`` (reference util/nethook.py:375-383), tolerant of the ``text_model.`` prefix.
    The path is walked directly first: building the name table costs ~0.2 ms on a CLIP text encoder and an edit call
    resolves a dozen names."""
    for cand in _candidates(name):
        m = _walk(model, cand)
        if m is not None:
            return m
    table = dict(model.named_modules())
    for cand in _candidates(name):
        if cand in table:
            return table[cand]
    raise LookupError(name)


def get_parameter(model, name):
    """``dict(model.named_parameters())[name]`` (reference util/nethook.py:385-392), same prefix tolerance, same fast path."""
    for cand in _candidates(name):
        owner, _, leaf = cand.rpartition(".")
        m = _walk(model, owner)
        if m is not None:
            p = getattr(m, "_parameters", {}).get(leaf)
            if p is not None:
                return p
    table = dict(model.named_parameters())
    for cand in _candidates(name):
        if cand in table:
            return table[cand]
    raise LookupError(name)


def set_requires_grad(requires_grad, *models):
    for m in models:
        if isinstance(m, torch.nn.Module):
            for p in m.parameters():
                p.requires_grad = requires_grad
        elif isinstance(m, (torch.nn.Parameter, torch.Tensor)):
            m.requires_grad = requires_grad
        else:
            raise TypeError(f"unknown type {type(m)!r}")


def _keep(x, clone, detach):
    if isinstance(x, torch.Tensor):
        if detach:
            x = x.detach()
        if clone:
            x = x.clone()
        return x
    if isinstance(x, (tuple, list)):
        return type(x)(_keep(v, clone, detach) for v in x)
    if isinstance(x, dict):
        return type(x)((k, _keep(v, clone, detach)) for k, v in x.items())
    return x


class Trace(contextlib.AbstractContextManager):
    """Retains the input and/or output of one named sub-module during a forward pass.

    ``edit_output(output, layer)`` may replace the output; ``stop=True`` aborts the forward right
    after the layer ran (the abort is swallowed on exit of the ``with`` block)."""

    def __init__(self, module, layer=None, retain_output=True, retain_input=False, clone=False, detach=False,
                 edit_output=None, stop=False):
        self.layer = layer
        self.stop = stop
        self.input = self.output = None
        target = get_module(module, layer) if layer is not None else module

        def hook(mod, inputs, output):
            if retain_input:
                self.input = _keep(inputs[0] if len(inputs) == 1 else inputs, clone, detach)
            if edit_output is not None:
                output = edit_output(output, self.layer)
            if retain_output:
                self.output = _keep(output, clone, detach)
            if stop:
                raise StopForward()
            return output

        self._handle = target.register_forward_hook(hook)

    def __exit__(self, exc_type, exc, tb):
        self.close()
        return bool(self.stop and exc_type is not None and issubclass(exc_type, StopForward))

    def close(self):
        self._handle.remove()


class TraceDict(OrderedDict, contextlib.AbstractContextManager):
    """``Trace`` over several layers at once; ``td[layer].input`` / ``.output``.  With ``stop=True``
    the forward is abandoned after the LAST listed layer has run."""

    def __init__(self, module, layers=None, retain_output=True, retain_input=False, clone=False, detach=False,
                 edit_output=None, stop=False):
        super().__init__()
        self.stop = stop
        layers = list(layers or [])
        for i, layer in enumerate(layers):
            self[layer] = Trace(module, layer, retain_output=retain_output, retain_input=retain_input, clone=clone,
                                detach=detach, edit_output=edit_output, stop=stop and i == len(layers) - 1)

    def __exit__(self, exc_type, exc, tb):
        self.close()
        return bool(self.stop and exc_type is not None and issubclass(exc_type, StopForward))

    def close(self):
        for tr in reversed(list(self.values())):
            tr.close()
