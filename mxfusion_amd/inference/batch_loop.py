"""BatchInferenceLoop (mxfusion/inference/batch_loop.py:19-61) and its data-parallel form.

`param_dict` is the InferenceParameters object: one flat leaf, one fused Adam kernel (mxf_adam_step).  The
multi-GPU loop shards the Monte-Carlo samples over ranks (one process per GPU) and sums the flat gradient with
ONE RCCL all-reduce per step over xGMI -- the reference has no counterpart (single ctx)."""
import torch

from .. import ops
from .grad_loop import GradLoop


class _Adam(object):
    """gluon.Trainer.step(batch_size) for the optimizers the reference's loops are called with by name (batch_loop.py:29-44 passes
    `optimizer` and {'learning_rate': lr} to the Trainer): MXNet 'adam' (the default), 'sgd' (plain, or
    optimizer=('sgd', {'momentum': 0.9, 'wd': 0.0})), 'nag', 'rmsprop' (non-centred), 'adagrad', 'adadelta' -- options as a
    (name, {...}) pair with MXNet's keyword names; rescale_grad = 1/batch_size.  One fused kernel over the flat parameter buffer."""

    # MXNet 1.x defaults of the rules served by mxf_opt_step: (first parameter, its keyword, epsilon)
    _RULES = {'rmsprop': (0.9, 'gamma1', 1e-8), 'adagrad': (0.0, None, 1e-7), 'adadelta': (0.9, 'rho', 1e-5), 'nag': (0.0, 'momentum', 0.0)}

    def __init__(self, params, learning_rate, optimizer='adam'):
        opts = {}
        if isinstance(optimizer, (tuple, list)):
            optimizer, opts = optimizer[0], dict(optimizer[1])
        if optimizer not in ('adam', 'sgd') and optimizer not in self._RULES:
            raise NotImplementedError("optimizer %r: 'adam' (the reference default), 'sgd', 'nag', 'rmsprop', 'adagrad' and 'adadelta' are implemented"
                                      % (optimizer,))
        if optimizer == 'rmsprop' and opts.get('centered', False):
            raise NotImplementedError("optimizer 'rmsprop': the centred variant is not implemented")
        self.kind, self.opts = optimizer, opts
        self.params, self.lr, self.t = params, learning_rate, 0
        flat = params.flat.detach()
        self.v = None
        if optimizer == 'adam':
            self.m, self.v = torch.zeros_like(flat), torch.zeros_like(flat)
        elif optimizer == 'sgd':
            self.m = torch.zeros_like(flat) if opts.get('momentum', 0.0) != 0.0 else None
        else:
            self.m = torch.zeros_like(flat)
            if optimizer == 'adadelta':
                self.v = torch.zeros_like(flat)

    def step(self, batch_size=1):
        self.t += 1
        flat = self.params.flat
        if self.kind == 'adam':
            ops.adam_step_(flat.detach(), flat.grad, self.m, self.v, self.lr, self.t, beta1=self.opts.get('beta1', 0.9),
                           beta2=self.opts.get('beta2', 0.999), epsilon=self.opts.get('epsilon', 1e-8), rescale_grad=1.0 / batch_size)
        elif self.kind == 'sgd':
            ops.sgd_step_(flat.detach(), flat.grad, self.m, self.lr, momentum=self.opts.get('momentum', 0.0), wd=self.opts.get('wd', 0.0),
                          rescale_grad=1.0 / batch_size)
        else:
            p1, key, eps = self._RULES[self.kind]
            ops.opt_step_(self.kind, flat.detach(), flat.grad, self.m, self.v, self.lr, self.opts.get(key, p1) if key else 0.0,
                          self.opts.get('epsilon', self.opts.get('eps', eps)), wd=self.opts.get('wd', 0.0), rescale_grad=1.0 / batch_size)
        self.params.zero_grad()


class _GraphStepMixin(object):
    """Replay of one step's forward + reverse pass as a hipGraph (torch.cuda.CUDAGraph): shared by the batch loop and (r05) the minibatch loop,
    whose small steps -- the reference's svgp_regression notebook runs minibatches of 10 rows with 20 inducing points -- are paced by the host's
    ~280 launches per step (1.03 ms eager against 0.42 ms replayed, tests/probes/small_step.py)."""

    def _graph_step(self, infr_executor, data, param_dict):
        from .. import _lib
        dev = param_dict.flat.device.index if param_dict.flat.device.index is not None else torch.cuda.current_device()
        key = (id(infr_executor), tuple((tuple(d.shape), d.dtype) for d in data))
        st = getattr(self, '_gstate', None)
        if st is None or st.get('flat') is not param_dict.flat or st.get('key') != key:
            if st is not None and 'graph' in st:
                torch.cuda.synchronize()         # (never destroy a graph that may still be executing)
            st = self._gstate = {'n': 0, 'flat': param_dict.flat, 'key': key}
        from ..modules.gp_modules._fused import Float32Guard
        if 'graph' in st and st.get('f32_epoch') != Float32Guard.poll_all(dev):
            # an SVGP module's float32 guard changed its level (condition number of Kuu crossed a limit): the captured launches carry the OLD
            # form -- drop the graph, warm up once eagerly (the modules now pick the new form), capture again
            torch.cuda.synchronize()             # (the last replay may still be running: destroying an executing hipGraphExec crashed the
            for k in ('graph', 'loss', 'grad', 'data', 'ws_gen', 'f32_epoch'):     #  HIP runtime's own thread -- r05, flaky segfault in the full suite)
                st.pop(k, None)
            st['n'] = 1
        if 'graph' in st and st['ws_gen'] != _lib.workspace_generation(dev):
            # the library re-allocated its scratch since the capture (a larger call in between: a prediction, another module): the
            # captured kernels carry the OLD scratch addresses -- drop the graph, warm up once more eagerly, capture again
            torch.cuda.synchronize()
            for k in ('graph', 'loss', 'grad', 'data', 'ws_gen'):
                st.pop(k, None)
            st['n'] = 1
        if 'graph' not in st:
            if st['n'] < 2:                      # eager warm-up, on a side stream (the documented whole-step capture recipe: autograd's
                st['n'] += 1                     # AccumulateGrad node of the flat leaf must not be bound to the default stream)
                side = st.setdefault('stream', torch.cuda.Stream())
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    loss, loss_for_gradient = infr_executor(*data)
                    loss_for_gradient.backward()
                torch.cuda.current_stream().wait_stream(side)
                return self._exchange(param_dict, loss.detach())
            torch.cuda.synchronize()
            param_dict.zero_grad()
            gen0 = _lib.workspace_generation(dev)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                loss, loss_for_gradient = infr_executor(*data)
                loss_for_gradient.backward()
            if _lib.workspace_generation(dev) != gen0:      # the library never allocates inside a capture; belt and braces
                raise RuntimeError('mxfusion_amd: the scratch workspace was re-allocated during hipGraph capture')
            st.update(graph=g, loss=loss.detach(), grad=param_dict.flat.grad, data=[d for d in data], ws_gen=gen0, f32_epoch=Float32Guard.epoch)
        for d, d0 in zip(data, st['data']):
            if d is not d0:
                d0.copy_(d)
        st['graph'].replay()
        param_dict.flat.grad = st['grad']
        return self._exchange(param_dict, st['loss'])


class BatchInferenceLoop(GradLoop, _GraphStepMixin):
    """use_graph=True: after two eager warm-up steps (scratch growth, lazy initialisation) the forward + reverse pass of one step
    (~280 kernel launches on three streams) is captured ONCE into a hipGraph (torch.cuda.CUDAGraph) and replayed; the gradient
    exchange and the Adam kernel stay outside the graph.  The data tensors passed to step() must then be the same objects every
    step (their contents may change)."""

    def __init__(self, use_graph=False):
        self.use_graph = use_graph
        self._gstate = None

    def _make_trainer(self, param_dict, learning_rate, optimizer):
        """gluon.Trainer seam (batch_loop.py:29-44): the fused HIP optimiser over the flat buffer; CPU tests of the loop swap it."""
        return _Adam(param_dict, learning_rate, optimizer)

    def run(self, infr_executor, data, param_dict, ctx, optimizer='adam', learning_rate=1e-3, max_iter=1000, n_prints=10, verbose=False):
        trainer = self._make_trainer(param_dict, learning_rate, optimizer)
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
            infr_executor(*self._local(data))

    def step(self, infr_executor, data, param_dict):
        """record -> forward -> backward (batch_loop.py:52-54) + the gradient exchange hook; returns the loss."""
        if getattr(self, 'use_graph', False) and param_dict.flat.is_cuda:
            return self._graph_step(infr_executor, data, param_dict)
        loss, loss_for_gradient = infr_executor(*data)
        loss_for_gradient.backward()
        return self._exchange(param_dict, loss)

    def _exchange(self, param_dict, loss):
        """Exchange hook between backward and the optimiser step; returns the job's loss (one GPU: nothing to do)."""
        return loss

    def _local(self, data):
        return data


class _OneCollectiveExchange(object):
    """ONE collective per step, carrying 'flat gradient + scalar loss'.  The local gradient (weighted 1 / world when the
    samples are sharded: the objective is a mean over samples; weight 1 when the rows are: the ranks' objectives add up) is moved into a
    persistent exchange buffer of numel + 2 elements by the same kernel that applies the weight, the local loss goes into the two tail
    slots (value in the buffer's dtype + the remainder, so that a float64 loss survives a float32 gradient buffer), ONE all-reduce(sum)
    runs over the buffer, and the reduced gradient is handed to the optimiser as a VIEW of the buffer -- no copy back, no second
    latency-bound collective on a 2.4-4.5 ms step.  `collectives` counts the all-reduces issued (the tests assert one per step)."""

    collectives = 0

    def _world(self):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_world_size(self.process_group)
        return 1

    def _rank(self):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(self.process_group)
        return 0

    def _sync_parameters(self, param_dict):
        """Replicas must start from identical parameters (un-set ones are drawn from the host RNG): once, before the first step."""
        if not getattr(self, '_synced', False) and self._world() > 1:
            import torch.distributed as dist
            with torch.no_grad():
                dist.broadcast(param_dict.flat.data, src=0, group=self.process_group)
        self._synced = True

    def _exchange(self, param_dict, loss):
        world = self._world()
        if world <= 1:
            return loss
        import torch.distributed as dist
        flat = param_dict.flat
        g = flat.grad
        n = g.numel()
        buf = getattr(self, '_xbuf', None)
        if buf is None or buf.numel() != n + 2 or buf.dtype != g.dtype or buf.device != g.device:
            buf = self._xbuf = torch.empty(n + 2, dtype=g.dtype, device=g.device)
        w = 1.0 / world if self.shard == 'samples' else 1.0
        head = buf[:n]
        if g.data_ptr() != head.data_ptr():
            torch.mul(g.detach().reshape(-1), w, out=head)           # weight + move in one pass over the gradient
        elif w != 1.0:
            head.mul_(w)
        l = loss.detach().reshape(1)
        if l.dtype == buf.dtype:
            torch.mul(l, w, out=buf[n:n + 1])
            buf[n + 1:].zero_()
        else:                                                        # e.g. float64 objective over a float32 parameter buffer
            hi = (l * w).to(buf.dtype)
            buf[n:n + 1].copy_(hi)
            buf[n + 1:].copy_((l * w - hi.to(l.dtype)).to(buf.dtype))
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.process_group)
        self.collectives += 1
        if getattr(param_dict, '_train_flat', None) is not None:     # GradTransferInference: the callers' tensors hold views of THIS gradient
            g.copy_(head.view_as(g))
        else:
            flat.grad = head.view_as(flat)
        return (buf[n].to(l.dtype) + buf[n + 1].to(l.dtype)).reshape(loss.shape)


class DistributedBatchInferenceLoop(_OneCollectiveExchange, BatchInferenceLoop):
    """Data-parallel batch loop (backend 'nccl' is RCCL on ROCm; tests use 'gloo', on CPU tensors here and on device tensors on the GPU box):
    ONE all-reduce per step carries the flat gradient and the loss (_OneCollectiveExchange).

    shard='samples': every rank evaluates its shard of the MC samples (the inference algorithm's num_samples is the LOCAL count); objective
      and gradient are the mean over ranks.
    shard='rows' (models without a sample axis; `row_variables` = the variables whose factors are sums over data rows, e.g. [m.Y] of an SVGP
      model): every rank evaluates rows [r N / world, (r + 1) N / world) of the data -- every data tensor whose leading dimension is the row
      count is split (`global_variables` names observed variables to leave whole) --, the row-independent factors carry weight 1 / world (see DistributedMinibatchInferenceLoop); gradient and loss are summed."""

    def __init__(self, process_group=None, use_graph=False, shard='samples', row_variables=None, global_variables=None):
        super(DistributedBatchInferenceLoop, self).__init__(use_graph=use_graph)
        if shard not in ('samples', 'rows'):
            raise ValueError("shard must be 'samples' or 'rows'")
        if shard == 'rows' and not row_variables:
            raise ValueError("shard='rows' needs row_variables: the variables whose factors are sums over data rows")
        self.process_group = process_group
        self.shard = shard
        self.rv_scaling = {v.uuid: 1.0 for v in row_variables} if shard == 'rows' else None
        self.global_uuids = {v.uuid for v in (global_variables or [])}      # observed variables that are NOT per-row even if their leading size is N
        self._local_cache = None

    def global_weight(self):
        return 1.0 / self._world() if self.shard == 'rows' else None

    def bind_data(self, observed_uuids):
        """GradBasedInference.run tells the loop which observed variable each data tensor belongs to (same order as `data`)."""
        self._data_uuids = list(observed_uuids)

    def _local(self, data):
        """This rank's rows of the data (row sharding); the slices are cached per data object so that a captured graph keeps its inputs.

        The row count N is the leading dimension of a `row_variables` tensor (when the loop knows which tensor belongs to which variable:
        bind_data; otherwise the largest leading dimension).  Every data tensor with N leading rows is split -- the inputs that go with the
        rows -- except the variables named in `global_variables` (e.g. observed inducing inputs with M == N).  N < world refuses."""
        world = self._world()
        if self.shard != 'rows' or world <= 1:
            return data
        import torch.distributed as dist
        key = tuple(id(d) for d in data)
        if self._local_cache is None or self._local_cache[0] != key:
            r = dist.get_rank(self.process_group)
            uuids = getattr(self, '_data_uuids', None)
            if uuids is not None and len(uuids) != len(data):
                uuids = None
            has_rows = lambda d: hasattr(d, 'shape') and d.dim() > 0
            n = None
            if uuids is not None:
                rows = [d.shape[0] for u, d in zip(uuids, data) if u in self.rv_scaling and has_rows(d)]
                if rows:
                    if len(set(rows)) != 1:
                        raise ValueError("shard='rows': the row_variables disagree on the number of rows (%s)" % sorted(set(rows)))
                    n = rows[0]
            if n is None:
                n = max(d.shape[0] for d in data if has_rows(d))
            if n < world:
                raise ValueError("shard='rows': %d data rows cannot be shared among %d ranks" % (n, world))
            keep = self.global_uuids
            loc = [torch.tensor_split(d, world)[r] if (has_rows(d) and d.shape[0] == n and not (uuids is not None and uuids[i] in keep)) else d
                   for i, d in enumerate(data)]
            self._local_cache = (key, loc, list(data))         # (the data list is kept alive: ids are only unique among live objects)
        return self._local_cache[1]

    def step(self, infr_executor, data, param_dict):
        self._sync_parameters(param_dict)
        return super(DistributedBatchInferenceLoop, self).step(infr_executor, self._local(data), param_dict)
