"""HF-named per-tensor parameters over flat storage (``named_parameters()`` names containing ``layer.{i}``).

The kernels read two flat fp32 tensors (``flat_decay`` / ``flat_nodecay``); the reference's own code iterates per-tensor
``nn.Parameter``s by HF name:

  * ``Lamb(model.parameters())`` with one trust ratio per tensor           ANCE/drivers/run_ann.py:128-147, ANCE/utils/lamb.py:71-121
  * ``iDROLoss._params``: parameters whose name contains ``layer.9`` ...   ANCE/model/dro_loss.py:174-190
  * ``clip_grad_norm_(model.parameters(), max_grad_norm)``                 ANCE/drivers/run_ann.py:345-347
  * HF Trainer's decay grouping: no decay on ``bias`` names and on parameters of ``nn.LayerNorm`` modules
                                                                           COCO/trainer.py:66-70 -> Trainer.create_optimizer

So the module carries a SHELL tree of ``nn.Linear`` / ``nn.LayerNorm`` / ``nn.Embedding`` objects under the HF names whose
``weight`` / ``bias`` are ``ViewParameter``s: aliases of slices of the flat tensors (an in-place optimizer update of a view
updates the storage the kernels read), whose ``.grad`` is - on every access - the matching slice of the flat gradient that the
native backward wrote.  Nothing is copied in either direction, and a step that never touches the views (``param_groups()`` +
``FlatAdamW`` / ``FlatLamb``, the fast path) pays nothing for them.  The flat tensors themselves are NOT registered
parameters any more: ``parameters()`` yields each weight exactly once.

Limits (stated in INTEGRATION.md): the shells never run (the forward is native); ``requires_grad`` is all-or-nothing per
module (``module.requires_grad_(False)`` works, freezing single tensors does not); torch DDP cannot wrap the module (the
views never receive autograd gradients) - use ``enable_grad_allreduce``.
"""
from __future__ import annotations

import copy
import weakref
from collections import OrderedDict
from typing import List

import torch
from torch import nn

__all__ = ["ViewParameter", "FlatParamsMixin"]

FLAT_NAMES = ("flat_decay", "flat_nodecay")


class ViewParameter(nn.Parameter):
    """An ``nn.Parameter`` that aliases ``flat[off : off + n].view(shape)``; ``.grad`` aliases the same slice of ``flat.grad``."""

    def __new__(cls, data, owner=None, which=0, index=0, off=0, requires_grad=True):
        t = torch.Tensor._make_subclass(cls, data, requires_grad)
        t._owner = weakref.ref(owner) if owner is not None else None
        t._which, t._index, t._off = which, index, off
        return t

    def _flat(self):
        owner = self._owner() if self._owner is not None else None
        return (owner, owner.__dict__[FLAT_NAMES[self._which]]) if owner is not None else (None, None)

    @property
    def grad(self):
        owner, flat = self._flat()
        if flat is None or flat.grad is None or self._index in owner._grad_reset[self._which]:
            return None
        return flat.grad[self._off:self._off + self.numel()].view(self.shape)

    @grad.setter
    def grad(self, value):
        owner, flat = self._flat()
        if flat is None:
            return
        reset = owner._grad_reset[self._which]
        if value is None:  # optimizer.zero_grad() / module.zero_grad(): once every view of the flat is cleared, so is the flat
            reset.add(self._index)
            if len(reset) == owner._n_views[self._which]:
                flat.grad = None
                reset.clear()
            return
        if flat.grad is None:
            flat.grad = torch.zeros_like(flat.data)
            reset.update(i for i in range(owner._n_views[self._which]) if i != self._index)
        reset.discard(self._index)
        flat.grad[self._off:self._off + self.numel()].view(self.shape).copy_(value)

    # ``p.data`` (the reference's Lamb reads and writes ``p.data``, ANCE/utils/lamb.py:97-120): what is handed out is a detached
    # alias that SHARES the version counter of the flat's view base (every view derives from that one alias), so an in-place write
    # through it - now or through an alias a helper cached across steps (EMA / SWA, an optimizer keeping ``p.data`` in its state) -
    # is seen by ``_params_version`` and refreshes the bf16 weight shadow; merely reading ``p.data`` (logging, norms) costs nothing.
    @property
    def data(self):
        return self.detach()

    @data.setter
    def data(self, value):  # keep aliasing the flat: copy instead of re-binding the storage
        with torch.no_grad():
            self.detach().copy_(value)

    def __deepcopy__(self, memo):  # a detached plain parameter (deep copies of the owning module rebuild their own views)
        return nn.Parameter(self.detach().clone(), self.requires_grad)

    def __reduce_ex__(self, proto):
        return (nn.Parameter, (self.detach().clone(), self.requires_grad))


class _Shell(nn.Module):
    """container node of the HF module tree (``encoder``, ``layer``, ``0``, ``attention`` ...); never runs"""

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("this module only carries HF-named parameter views; the forward pass is native (CocoBertModel.forward)")


class _LinearShell(nn.Linear):
    def __init__(self):
        nn.Module.__init__(self)
        self.in_features = self.out_features = 0


class _LayerNormShell(nn.LayerNorm):  # isinstance(m, nn.LayerNorm): what HF's get_parameter_names excludes from weight decay
    def __init__(self, eps):
        nn.Module.__init__(self)
        self.normalized_shape, self.eps, self.elementwise_affine = (), eps, True


class _EmbeddingShell(nn.Embedding):
    def __init__(self):
        nn.Module.__init__(self)
        self.num_embeddings = self.embedding_dim = 0
        self.padding_idx = self.max_norm = None
        self.norm_type, self.scale_grad_by_freq, self.sparse = 2.0, False, False


class FlatParamsMixin:
    """For an ``nn.Module`` with ``self.layout`` (a ``_Layout``: HF name -> (flat index, offset, shape)) and the two flat
    tensors.  Call ``_set_flat`` for both flats, then ``_build_views()``."""

    # ---------------------------------------------------------------- the flats: plain attributes, not registered parameters
    def __setattr__(self, name, value):
        if name in FLAT_NAMES:
            return self._set_flat(name, value)
        return super().__setattr__(name, value)

    def _set_flat(self, name: str, value) -> None:
        if not isinstance(value, nn.Parameter):
            value = nn.Parameter(value)
        self._parameters.pop(name, None)
        self.__dict__[name] = value
        which = FLAT_NAMES.index(name)
        self.__dict__.setdefault("_grad_reset", [set(), set()])[which] = set()
        self.__dict__.setdefault("_vbase", [None, None])[which] = None
        value.register_hook(lambda g, self_ref=weakref.ref(self), w=which: FlatParamsMixin._before_accumulate(self_ref, w))

    @staticmethod
    def _before_accumulate(self_ref, which):
        """tensor hook of a flat (runs before autograd adds the incoming gradient): views cleared one by one since the last
        backward (``p.grad = None`` on some, not all of them) must read as zero + the new gradient"""
        self = self_ref()
        if self is None:
            return None
        reset = self._grad_reset[which]
        if reset:
            flat = self.__dict__[FLAT_NAMES[which]]
            if flat.grad is not None:
                for v in self._views[which]:
                    if v._index in reset:
                        flat.grad[v._off:v._off + v.numel()].zero_()
            reset.clear()
        return None

    # ---------------------------------------------------------------- the HF-named shell tree
    def _build_views(self) -> None:
        """(re)create the shell modules and their ViewParameters from ``self.layout`` and the current flat storage"""
        for root in self.__dict__.get("_shell_roots", ()):
            self._modules.pop(root, None)
            self._parameters.pop(root, None)
        flats = [self.__dict__[n] for n in FLAT_NAMES]
        self.__dict__["_vbase"] = [f.data for f in flats]  # every view of a flat derives from ONE alias: shared version counter
        views: List[List[ViewParameter]] = [[], []]
        roots = []
        eps = getattr(getattr(self, "config", None), "layer_norm_eps", 1e-12)
        for name, (which, off, shape) in self.layout.names.items():
            n = 1
            for s in shape:
                n *= s
            vp = ViewParameter(self._vbase[which][off:off + n].view(shape), self, which, len(views[which]), off,
                               requires_grad=flats[which].requires_grad)
            views[which].append(vp)
            parts = name.split(".")
            mod = self
            for depth, part in enumerate(parts[:-1]):
                last = depth == len(parts) - 2
                child = mod._modules.get(part)
                if child is None:
                    if not last:
                        child = _Shell()
                    elif part == "LayerNorm":
                        child = _LayerNormShell(eps)
                    elif part.endswith("_embeddings"):
                        child = _EmbeddingShell()
                    elif len(self.layout.names.get(".".join(parts[:-1]) + ".weight", (0, 0, ()))[2]) == 2:
                        child = _LinearShell()
                    else:
                        child = _Shell()
                    nn.Module.add_module(mod, part, child)
                    if mod is self:
                        roots.append(part)
                mod = child
            mod.register_parameter(parts[-1], vp)
            if isinstance(mod, _LinearShell) and len(shape) == 2:
                mod.out_features, mod.in_features = shape
            elif isinstance(mod, _LayerNormShell):
                mod.normalized_shape = tuple(shape)
            elif isinstance(mod, _EmbeddingShell):
                mod.num_embeddings, mod.embedding_dim = shape
        self.__dict__["_views"] = views
        self.__dict__["_n_views"] = [len(views[0]), len(views[1])]
        self.__dict__["_shell_roots"] = tuple(roots)
        self.__dict__["_grad_reset"] = [set(), set()]

    def _params_version(self):
        """changes whenever the flat storage was written in place - through a flat (``FlatAdamW``, ``load_state_dict``), through
        any view (a per-tensor torch optimizer) or through an alias ``view.data`` handed out at any time"""
        fd = self.__dict__[FLAT_NAMES[0]]
        vb = self.__dict__.get("_vbase", (None, None))[0]
        return (fd._version, vb._version if vb is not None else -1)

    def _shadow_stale(self) -> bool:
        return self._shadow_version != self._params_version()

    def _shadow_mark_fresh(self) -> None:
        self._shadow_version = self._params_version()

    # ---------------------------------------------------------------- nn.Module plumbing that must see the flats
    def _apply(self, fn, recurse=True):
        for name in FLAT_NAMES:
            p = self.__dict__[name]
            with torch.no_grad():
                p.data = fn(p.data)
                if p.grad is not None:
                    p.grad.data = fn(p.grad.data)
        for key, buf in self._buffers.items():
            if buf is not None:
                self._buffers[key] = fn(buf)
        for key, m in self._modules.items():
            if m is not None and key not in self.__dict__.get("_shell_roots", ()):
                m._apply(fn)
        if hasattr(self, "_shadow"):
            self._shadow, self._shadow_version = None, -1
        self._build_views()
        return self

    def zero_grad(self, set_to_none: bool = True) -> None:
        for name in FLAT_NAMES:
            p = self.__dict__[name]
            if p.grad is not None:
                if set_to_none:
                    p.grad = None
                else:
                    p.grad.detach_().zero_()
        for r in self._grad_reset:
            r.clear()
        for key, m in self._modules.items():
            if m is not None and key not in self._shell_roots:
                m.zero_grad(set_to_none)

    def requires_grad_(self, requires_grad: bool = True):
        for name in FLAT_NAMES:
            self.__dict__[name].requires_grad_(requires_grad)
        return super().requires_grad_(requires_grad)

    def flat_parameters(self):
        """the two flat tensors (what ``param_groups()`` hands to the fused optimizers)"""
        return [self.__dict__[n] for n in FLAT_NAMES]

    def __deepcopy__(self, memo):
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        shells = set(self.__dict__.get("_shell_roots", ()))
        transient = {"_views", "_vbase", "_grad_reset", "_dp_hooks", "_shadow"}
        for k, v in self.__dict__.items():
            if k == "_modules":
                new.__dict__[k] = OrderedDict((n, copy.deepcopy(m, memo)) for n, m in v.items() if n not in shells)
            elif k in FLAT_NAMES:
                continue
            elif k in transient:
                if k not in ("_views", "_vbase", "_grad_reset"):  # (those three are rebuilt below)
                    new.__dict__[k] = None
            else:
                new.__dict__[k] = copy.deepcopy(v, memo)
        new.__dict__["_shell_roots"] = ()
        for name in FLAT_NAMES:
            src = self.__dict__[name]
            p = nn.Parameter(src.data.clone(), src.requires_grad)
            if src.grad is not None:
                p.grad = src.grad.clone()
            new._set_flat(name, p)
        if "_shadow_version" in new.__dict__:
            new.__dict__["_shadow_version"] = -1
        new._build_views()
        return new
