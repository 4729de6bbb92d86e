"""MinibatchInferenceLoop (mxfusion/inference/minibatch_loop.py:21-95): shuffled minibatches with
last_batch='rollover', rv_scaling = N/B pushed into the factors as log_pdf_scaling, Trainer.step(batch_size=B)
(=> gradient / B) -- and its data-parallel form (BASELINE.json configs[3]: minibatches with the
Monte-Carlo samples sharded over the GPUs of a node, one RCCL all-reduce of the flat gradient per minibatch)."""
import torch

from .batch_loop import _Adam, _GraphStepMixin, _OneCollectiveExchange
from .grad_loop import GradLoop


class MinibatchInferenceLoop(GradLoop, _GraphStepMixin):
    """use_graph=True (r05; no reference counterpart): the forward + reverse pass of a minibatch step is captured once into a hipGraph and
    replayed -- every minibatch is gathered into the SAME device buffers first (one index kernel per data tensor), the gradient exchange and
    the optimiser stay outside the graph.  For small models the step is paced by the host's launches, not by the device: the reference's
    svgp_regression notebook shape (N = 1000, 20 inducing points, minibatches of 10) takes 1.03 ms per step eager and 0.42 ms replayed.
    Large steps (thousands of rows x 1024 inducing points) are device-bound and replay no faster: leave it off there."""

    def __init__(self, batch_size=100, rv_scaling=None, use_graph=False):
        super(MinibatchInferenceLoop, self).__init__()
        self.batch_size = batch_size
        self.rv_scaling = {v.uuid: s for v, s in rv_scaling.items()} if rv_scaling is not None else rv_scaling
        self.use_graph = use_graph
        self._gstate = None
        self._static = None

    # ---- seams ---------------------------------------------------------------------------------------------------------------------
    def _make_trainer(self, param_dict, learning_rate, optimizer):
        return _Adam(param_dict, learning_rate, optimizer)

    def _next_permutation(self, N, device, generator, permutations):
        """One epoch's shuffle (the reference's DataLoader(shuffle=True), minibatch_loop.py:68-70).  `permutations`: an iterator of index
        sequences injected by the caller (tests; the analogue of the rand_gen seam), else torch.randperm with `generator`."""
        if permutations is not None:
            return torch.as_tensor(next(permutations), dtype=torch.long).to(device)
        return torch.randperm(N, device=device, generator=generator)

    def _exchange(self, param_dict, loss):
        """Exchange hook between backward and the optimiser step; returns the job's loss (nothing to do on one GPU)."""
        return loss

    # ---- one minibatch -----------------------------------------------------------------------------------------------------------------
    def step(self, infr_executor, batch, param_dict, update_shape_constants=None):
        """record -> forward -> backward (minibatch_loop.py:78-82) + the exchange hook; returns the loss.  The caller owns the optimiser
        step (Trainer.step(batch_size=B), :86)."""
        if update_shape_constants is not None:
            update_shape_constants(batch)
        if getattr(self, 'use_graph', False) and param_dict.flat.is_cuda:
            st = self._static
            if st is None or len(st) != len(batch) or any(a.shape != b.shape or a.dtype != b.dtype for a, b in zip(st, batch)):
                st = self._static = [torch.empty_like(b) for b in batch]      # (a new shape re-captures: the graph state is keyed by the shapes)
            for a, b in zip(st, batch):
                a.copy_(b)
            return self._graph_step(infr_executor, st, param_dict)            # (runs the exchange hook itself)
        loss, loss_for_gradient = infr_executor(*batch)
        loss_for_gradient.backward()
        return self._exchange(param_dict, loss)

    def run(self, infr_executor, data, param_dict, ctx, optimizer='adam', learning_rate=1e-3, max_iter=1000, verbose=False,
            update_shape_constants=None, generator=None, permutations=None):
        trainer = self._make_trainer(param_dict, learning_rate, optimizer)
        N = data[0].shape[0]
        B = self.batch_size
        perms = iter(permutations) if permutations is not None else None
        carry = torch.empty(0, dtype=torch.long, device=data[0].device)
        self.epoch_losses = []
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
                # the epoch loss is accumulated ON THE DEVICE: the reference's `L_e += loss.asscalar()` (minibatch_loop.py:92) blocks the host on
                # every minibatch -- with a 2.5 ms step that is a full drain of the launch queue per step; one read per epoch here
                L_e = loss.detach().clone() if n_batches == 0 else L_e + loss.detach()      # (clone: a replayed graph returns the SAME tensor every step)
                n_batches += 1
            carry = idx[n_full * B:]
            if n_batches:
                self.epoch_losses.append(L_e / n_batches)             # device scalars; float() them when (if) they are wanted
                if len(self.epoch_losses) % 512 == 0:                  # long runs: the old entries become host floats (one drain per 512 epochs)
                    self.epoch_losses[-512:-256] = [float(x) for x in self.epoch_losses[-512:-256]]
            if verbose and n_batches:
                print('epoch-loss: {} '.format(float(L_e) / n_batches))
        self._trainer = trainer


class DistributedMinibatchInferenceLoop(_OneCollectiveExchange, MinibatchInferenceLoop):
    """Data-parallel minibatch loop (one process per GPU; backend 'nccl' is RCCL on ROCm, tests use 'gloo').  Every rank holds the full data
    set and walks the SAME shuffles (rank 0 draws each epoch's permutation and broadcasts it); ONE all-reduce per minibatch carries the flat
    gradient AND the loss (_OneCollectiveExchange), then every rank takes the identical Trainer.step(batch_size=B).

    shard='samples' (BASELINE.json configs[3]): every rank evaluates the whole minibatch with ITS shard of the Monte-Carlo samples (the
      inference algorithm's num_samples is the local count); objective and gradient are the mean over ranks.
    shard='rows' (second axis -- for models WITHOUT a sample axis, e.g. MAP on observed inputs as in the reference's
      svgp_regression notebook, where sample sharding leaves 7 of 8 GPUs idle): every rank evaluates rows [r B / world, (r + 1) B / world) of
      the minibatch.  Factors of the variables named in rv_scaling are sums over rows (that is what makes them minibatch-able,
      minibatch_loop.py:42-63) and see the rank's rows with the usual N / B scaling; every other factor -- priors of global parameters and,
      inside an SVGP module, -KL(q(u) || p(u)) (svgp_regression.py:93-97) -- is evaluated by every rank with weight 1 / world
      (InferenceAlgorithm.prepare_executor(global_weight=...)).  The ranks' objectives then ADD UP to the single-process objective: gradient
      and loss are summed, not averaged.  Modules whose bound is not a sum over rows (exact GP, Titsias sparse GP) refuse.
    The reference has no counterpart (single ctx); 1 rank reproduces MinibatchInferenceLoop exactly."""

    def __init__(self, batch_size=100, rv_scaling=None, process_group=None, shard='samples', use_graph=False):
        super(DistributedMinibatchInferenceLoop, self).__init__(batch_size=batch_size, rv_scaling=rv_scaling, use_graph=use_graph)
        if shard not in ('samples', 'rows'):
            raise ValueError("shard must be 'samples' or 'rows'")
        if shard == 'rows' and not self.rv_scaling:
            raise ValueError("shard='rows' needs rv_scaling: it names the variables whose factors are sums over data rows")
        self.process_group = process_group
        self.shard = shard
        self._synced = False

    def global_weight(self):
        """Weight of the row-independent factors in this rank's objective (GradBasedInference.create_executor asks for it)."""
        return 1.0 / self._world() if self.shard == 'rows' else None

    def _next_permutation(self, N, device, generator, permutations):
        perm = super(DistributedMinibatchInferenceLoop, self)._next_permutation(N, device, generator, permutations)
        if self._world() > 1:
            import torch.distributed as dist
            dist.broadcast(perm, src=0, group=self.process_group)        # identical minibatches on every rank
        return perm

    def step(self, infr_executor, batch, param_dict, update_shape_constants=None):
        world = self._world()
        self._sync_parameters(param_dict)
        if self.shard == 'rows' and world > 1:
            r = self._rank()
            batch = [torch.tensor_split(d, world)[r] for d in batch]     # this rank's rows of the (identical) minibatch
        return super(DistributedMinibatchInferenceLoop, self).step(infr_executor, batch, param_dict, update_shape_constants)
