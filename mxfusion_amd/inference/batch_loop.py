"""BatchInferenceLoop (mxfusion/inference/batch_loop.py:19-61) and its data-parallel form.

`param_dict` is the InferenceParameters object: one flat leaf, one fused Adam kernel (mxf_adam_step).  The
multi-GPU loop shards the Monte-Carlo samples over ranks (one process per GPU) and sums the flat gradient with
ONE RCCL all-reduce per step over xGMI -- the reference has no counterpart (single ctx)."""
import torch

from .. import ops
from .grad_loop import GradLoop


class _Adam(object):
    """MXNet Adam as driven by gluon.Trainer.step(batch_size): rescale_grad = 1/batch_size."""

    def __init__(self, params, learning_rate, optimizer='adam'):
        if optimizer != 'adam':
            raise NotImplementedError("only optimizer='adam' (the reference default) is implemented")
        self.params, self.lr, self.t = params, learning_rate, 0
        self.m = torch.zeros_like(params.flat.detach())
        self.v = torch.zeros_like(params.flat.detach())

    def step(self, batch_size=1):
        self.t += 1
        flat = self.params.flat
        ops.adam_step_(flat.detach(), flat.grad, self.m, self.v, self.lr, self.t, rescale_grad=1.0 / batch_size)
        flat.grad = None


class BatchInferenceLoop(GradLoop):
    def run(self, infr_executor, data, param_dict, ctx, optimizer='adam', learning_rate=1e-3, max_iter=1000, n_prints=10, verbose=False):
        trainer = _Adam(param_dict, learning_rate, optimizer)
        iter_step = max(max_iter // n_prints, 1)
        for i in range(max_iter):
            loss = self.step(infr_executor, data, param_dict)
            if verbose:
                print('\rIteration {} loss: {}\t\t\t\t'.format(i + 1, float(loss.detach())), end='')
                if ((i + 1) % iter_step == 0 and i > 0) or i == max_iter - 1:
                    print()
            trainer.step(batch_size=1)
        self._trainer = trainer
        with torch.no_grad():                      # batch_loop.py:61: one extra forward, discarded
            infr_executor(*data)

    def step(self, infr_executor, data, param_dict):
        """record -> forward -> backward (batch_loop.py:52-54) + the gradient exchange hook; returns the loss."""
        loss, loss_for_gradient = infr_executor(*data)
        loss_for_gradient.backward()
        self._exchange(param_dict)
        return loss

    def _exchange(self, param_dict):
        pass


class DistributedBatchInferenceLoop(BatchInferenceLoop):
    """Data-parallel batch loop: every rank evaluates its shard of the MC samples (the inference algorithm's
    num_samples is the LOCAL count) with the loss weighted 1/world_size, then the flat gradient is summed with one
    all-reduce.  backend 'nccl' is RCCL on ROCm; tests use 'gloo' on CPU tensors."""

    def __init__(self, process_group=None):
        self.process_group = process_group

    def _exchange(self, param_dict):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.process_group) > 1:
            g = param_dict.flat.grad
            g.div_(dist.get_world_size(self.process_group))
            dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.process_group)
