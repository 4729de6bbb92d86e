"""InferenceParameters (mxfusion/inference/inference_parameters.py:30-252).

MI355X layout: every trainable parameter is a view into ONE flat device buffer (a single autograd leaf), so
  * the optimiser is one fused HIP kernel over the buffer (mxf_adam_step),
  * the data-parallel exchange is ONE RCCL all-reduce of the flat gradient,
and `params[var]` / `params[var] = value` keep the reference semantics (constrained values in, constrained
values out; stored unconstrained through var.transformation)."""
import numpy as np
import torch

from ..common import config
from ..components.variables.variable import Variable, VariableType
from ..util.inference import realize_shape


class InferenceParameters(object):
    def __init__(self, constants=None, dtype=None, context=None):
        self.dtype = config.torch_dtype(dtype)
        self.device = context if context is not None else config.get_default_device()
        self._constants = {}
        self._var_ties = {}
        if constants is not None:
            for k, v in constants.items():
                self._constants[k.uuid if isinstance(k, Variable) else k] = v
        self._slices = {}       # uuid -> (offset, numel, shape)
        self._flat = None       # the single trainable leaf
        self._fixed = {}        # uuid -> tensor: non-trainable stores (posterior caches written through SET_)
        self._vars = {}         # uuid -> Variable
        self._views = None
        self._train_flat = None  # GradTransferInference: the externally owned trainable tensors, re-homed into one flat buffer

    # ---- construction ---------------------------------------------------------------------------------
    @property
    def constants(self):
        return self._constants

    @property
    def var_ties(self):
        return self._var_ties

    def update_constants(self, constants):
        self._constants.update({(k.uuid if isinstance(k, Variable) else k): v for k, v in constants.items()})

    def _to_tensor(self, value):
        if isinstance(value, torch.Tensor):
            return value.to(device=self.device, dtype=self.dtype)
        return torch.as_tensor(np.asarray(value), dtype=self.dtype).to(self.device)

    def _initial_raw(self, var, shape, generator):
        if var.initial_value is not None:
            val = self._to_tensor(var.initial_value).reshape(shape) if int(np.prod(shape)) == int(np.prod(np.shape(var.initial_value)) or 1) \
                else self._to_tensor(var.initial_value).expand(shape).clone()
            if var.transformation is not None:
                val = var.transformation.inverseTransform(val)      # variable.py:236-245 initial_value_before_transformation
            return val
        # MXNet default initialiser Uniform(0.07); host RNG, seeded by the caller
        return (torch.rand(shape, dtype=self.dtype, generator=generator) * 0.14 - 0.07).to(self.device)

    def initialize_params(self, graphs, observed_uuid, seed=None, carry=None):
        """inference_parameters.py:63-90.  `carry`: {uuid: raw tensor} to inherit (TransferInference)."""
        gen = torch.Generator().manual_seed(seed if seed is not None else int(np.random.randint(0, 2**31 - 1)))
        params = []
        for g in graphs:
            for v in g.get_parameters():
                if v.uuid in observed_uuid or v.uuid in self._constants or v.uuid in self._vars:
                    continue
                self._vars[v.uuid] = v
                params.append(v)
            for v in g.get_constants():
                if v.uuid not in self._constants:
                    self._constants[v.uuid] = self._to_tensor(v.constant) if not isinstance(v.constant, (int, float)) else \
                        torch.full((1,), float(v.constant), dtype=self.dtype, device=self.device)
        shape_consts = {k: v for k, v in self._constants.items() if isinstance(v, (int, np.integer))}
        trainable, off = [], 0
        for v in params:
            if getattr(v, 'is_posterior_cache', False):
                if carry is not None and v.uuid in carry:
                    self._fixed[v.uuid] = carry[v.uuid]
                continue
            if carry is not None and v.uuid in carry:
                shape = tuple(carry[v.uuid].shape)
            else:
                shape = realize_shape(v.shape, shape_consts)
            n = int(np.prod(shape)) if len(shape) else 1
            self._slices[v.uuid] = (off, n, shape)
            trainable.append(v)
            off += n
        flat = torch.zeros(max(off, 1), dtype=self.dtype, device=self.device)
        for v in trainable:
            o, n, shape = self._slices[v.uuid]
            raw = carry[v.uuid] if (carry is not None and v.uuid in carry) else self._initial_raw(v, shape, gen)
            flat[o:o + n] = raw.reshape(-1).to(self.dtype)
        self._flat = flat.requires_grad_(True)
        self._views = None
        for k in list(self._constants):
            if isinstance(self._constants[k], (int, np.integer)):
                continue
        return self

    def fix_all(self):
        """inference_parameters.py:139-141 (grad_req = 'null' on every parameter): the inherited buffer stops being a trainable leaf."""
        self._flat = self._flat.detach().requires_grad_(False)

    def set_train_params(self, tensors):
        """GradTransferInference's `train_params` (grad_based_inference.py:124-130): tensors owned by the caller (e.g. the parameters of
        a torch.nn policy network).  They are re-homed as views of ONE flat buffer, their .grad as views of one flat gradient, so that
        the optimiser stays one mxf_adam_step launch and the data-parallel exchange one all-reduce; `flat` then names that buffer."""
        ts = list(tensors.values()) if isinstance(tensors, dict) else list(tensors)
        total = sum(t.numel() for t in ts)
        flat = torch.zeros(max(total, 1), dtype=self.dtype, device=self.device)
        grad = torch.zeros_like(flat)
        off = 0
        for t in ts:
            n = t.numel()
            flat[off:off + n] = t.detach().reshape(-1).to(device=self.device, dtype=self.dtype)
            t.data = flat[off:off + n].view(t.shape)
            t.grad = grad[off:off + n].view(t.shape)
            t.requires_grad_(True)
            off += n
        self._train_flat = flat.requires_grad_(True)
        self._train_flat.grad = grad
        self._train_tensors = ts

    def zero_grad(self):
        if self._train_flat is not None:
            self._train_flat.grad.zero_()
        else:
            self._flat.grad = None

    # ---- access ----------------------------------------------------------------------------------------------
    @property
    def flat(self):
        """The buffer the optimiser steps and the gradient exchange reduces."""
        return self._train_flat if self._train_flat is not None else self._flat

    def tensors(self):
        """uuid -> raw (unconstrained) tensor; trainable ones are views of the flat leaf (autograd-connected)."""
        # one split node instead of one slice node per parameter: its reverse mode is ONE concatenation into the flat gradient, where
        # per-parameter slices each cost a zero-fill + copy + add over the whole (8 MB) buffer -- ~36 launches per step
        items = sorted(self._slices.items(), key=lambda kv: kv[1][0])
        sizes = [n for _, (o, n, shape) in items]
        pad = self._flat.numel() - sum(sizes)
        parts = torch.split_with_sizes(self._flat, sizes + ([pad] if pad else []))
        out = {u: parts[i].view(shape) for i, (u, (o, n, shape)) in enumerate(items)}
        out.update(self._fixed)
        return out

    def raw(self, key):
        u = key.uuid if isinstance(key, Variable) else key
        if u in self._slices:
            o, n, shape = self._slices[u]
            return self._flat.detach()[o:o + n].view(shape)
        return self._fixed[u]

    def __contains__(self, key):
        u = key.uuid if isinstance(key, Variable) else key
        return u in self._slices or u in self._fixed

    def __getitem__(self, key):
        """Constrained value (inference_parameters.py:178-204)."""
        v = self._vars.get(key.uuid if isinstance(key, Variable) else key, key if isinstance(key, Variable) else None)
        val = self.raw(key)
        if v is not None and v.transformation is not None:
            with torch.no_grad():
                val = v.transformation.transform(val)
        return val

    def __setitem__(self, key, item):
        """inference_parameters.py:206-222: store the unconstrained value."""
        u = key.uuid if isinstance(key, Variable) else key
        v = key if isinstance(key, Variable) else self._vars.get(u)
        item = self._to_tensor(item)
        if v is not None and v.transformation is not None:
            item = v.transformation.inverseTransform(item)
        if u in self._slices:
            o, n, shape = self._slices[u]
            with torch.no_grad():
                self._flat[o:o + n] = item.reshape(-1)
        else:
            if v is not None:
                self._vars[u] = v
            self._fixed[u] = item.detach()

    def export_raw(self):
        out = {u: self.raw(u).clone() for u in self._slices}
        out.update({u: t for u, t in self._fixed.items()})
        return out
