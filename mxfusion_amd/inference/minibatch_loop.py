"""MinibatchInferenceLoop (mxfusion/inference/minibatch_loop.py:21-95): shuffled minibatches with
last_batch='rollover', rv_scaling = N/B pushed into the factors as log_pdf_scaling, Trainer.step(batch_size=B)
(=> gradient / B) -- and its data-parallel form (BASELINE.json configs[3]: minibatches with the
Monte-Carlo samples sharded over the GPUs of a node, one RCCL all-reduce of the flat gradient per minibatch)."""
import torch

from .batch_loop import _Adam
from .grad_loop import GradLoop


class MinibatchInferenceLoop(GradLoop):
    def __init__(self, batch_size=100, rv_scaling=None):
        super(MinibatchInferenceLoop, self).__init__()
        self.batch_size = batch_size
        self.rv_scaling = {v.uuid: s for v, s in rv_scaling.items()} if rv_scaling is not None else rv_scaling

    # ---- seams ---------------------------------------------------------------------------------------------------------------------
    def _make_trainer(self, param_dict, learning_rate, optimizer):
        return _Adam(param_dict, learning_rate, optimizer)

    def _next_permutation(self, N, device, generator, permutations):
        """One epoch's shuffle (the reference's DataLoader(shuffle=True), minibatch_loop.py:68-70).  `permutations`: an iterator of index
        sequences injected by the caller (tests; the analogue of the rand_gen seam), else torch.randperm with `generator`."""
        if permutations is not None:
            return torch.as_tensor(next(permutations), dtype=torch.long).to(device)
        return torch.randperm(N, device=device, generator=generator)

    def _exchange(self, param_dict):
        """Gradient exchange hook between backward and the optimiser step (nothing to do on one GPU)."""
        pass

    # ---- one minibatch -----------------------------------------------------------------------------------------------------------------
    def step(self, infr_executor, batch, param_dict, update_shape_constants=None):
        """record -> forward -> backward (minibatch_loop.py:78-82) + the exchange hook; returns the loss.  The caller owns the optimiser
        step (Trainer.step(batch_size=B), :86)."""
        if update_shape_constants is not None:
            update_shape_constants(batch)
        loss, loss_for_gradient = infr_executor(*batch)
        loss_for_gradient.backward()
        self._exchange(param_dict)
        return loss

    def run(self, infr_executor, data, param_dict, ctx, optimizer='adam', learning_rate=1e-3, max_iter=1000, verbose=False,
            update_shape_constants=None, generator=None, permutations=None):
        trainer = self._make_trainer(param_dict, learning_rate, optimizer)
        N = data[0].shape[0]
        B = self.batch_size
        perms = iter(permutations) if permutations is not None else None
        carry = torch.empty(0, dtype=torch.long, device=data[0].device)
        for e in range(max_iter):
            perm = self._next_permutation(N, data[0].device, generator, perms)
            idx = torch.cat([carry, perm])                       # 'rollover': the remainder opens the next epoch
            n_full = idx.numel() // B
            L_e, n_batches = 0., 0
            for i in range(n_full):
                sel = idx[i * B:(i + 1) * B]
                batch = [d[sel] for d in data]
                loss = self.step(infr_executor, batch, param_dict, update_shape_constants)
                if verbose:
                    print('\repoch {} Iteration {} loss: {}\t\t\t'.format(e + 1, i + 1, float(loss.detach())), end='')
                trainer.step(batch_size=B)
                L_e += float(loss.detach())
                n_batches += 1
            carry = idx[n_full * B:]
            if verbose and n_batches:
                print('epoch-loss: {} '.format(L_e / n_batches))
        self._trainer = trainer


class DistributedMinibatchInferenceLoop(MinibatchInferenceLoop):
    """Minibatches x Monte-Carlo-sample sharding (one process per GPU; backend 'nccl' is RCCL on ROCm, tests use 'gloo'):
      * every rank holds the full data set and walks the SAME shuffles (rank 0 draws each epoch's permutation and broadcasts it),
      * every rank evaluates the minibatch with ITS shard of the MC samples (the inference algorithm's num_samples is the local count),
        the loss weighted 1/world_size,
      * the flat gradient is summed with ONE all-reduce per minibatch, then every rank takes the identical Trainer.step(batch_size=B).
    The reference has no counterpart (single ctx); 1 rank reproduces MinibatchInferenceLoop exactly."""

    def __init__(self, batch_size=100, rv_scaling=None, process_group=None):
        super(DistributedMinibatchInferenceLoop, self).__init__(batch_size=batch_size, rv_scaling=rv_scaling)
        self.process_group = process_group
        self._synced = False

    def _world(self):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_world_size(self.process_group)
        return 1

    def _next_permutation(self, N, device, generator, permutations):
        perm = super(DistributedMinibatchInferenceLoop, self)._next_permutation(N, device, generator, permutations)
        if self._world() > 1:
            import torch.distributed as dist
            dist.broadcast(perm, src=0, group=self.process_group)        # identical minibatches on every rank
        return perm

    def step(self, infr_executor, batch, param_dict, update_shape_constants=None):
        if not self._synced and self._world() > 1:
            import torch.distributed as dist
            with torch.no_grad():           # replicas must start from identical parameters (un-set ones are drawn from the host RNG)
                dist.broadcast(param_dict.flat.data, src=0, group=self.process_group)
            self._synced = True
        return super(DistributedMinibatchInferenceLoop, self).step(infr_executor, batch, param_dict, update_shape_constants)

    def _exchange(self, param_dict):
        world = self._world()
        if world > 1:
            import torch.distributed as dist
            g = param_dict.flat.grad
            g.div_(world)
            dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.process_group)
