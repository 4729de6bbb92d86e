"""MinibatchInferenceLoop (mxfusion/inference/minibatch_loop.py:21-95): shuffled minibatches with
last_batch='rollover', rv_scaling = N/B pushed into the factors as log_pdf_scaling, Trainer.step(batch_size=B)
(=> gradient / B)."""
import torch

from .batch_loop import _Adam
from .grad_loop import GradLoop


class MinibatchInferenceLoop(GradLoop):
    def __init__(self, batch_size=100, rv_scaling=None):
        super(MinibatchInferenceLoop, self).__init__()
        self.batch_size = batch_size
        self.rv_scaling = {v.uuid: s for v, s in rv_scaling.items()} if rv_scaling is not None else rv_scaling

    def run(self, infr_executor, data, param_dict, ctx, optimizer='adam', learning_rate=1e-3, max_iter=1000, verbose=False,
            update_shape_constants=None, generator=None):
        trainer = _Adam(param_dict, learning_rate, optimizer)
        N = data[0].shape[0]
        B = self.batch_size
        carry = torch.empty(0, dtype=torch.long, device=data[0].device)
        for e in range(max_iter):
            perm = torch.randperm(N, device=data[0].device, generator=generator)
            idx = torch.cat([carry, perm])                       # 'rollover': the remainder opens the next epoch
            n_full = idx.numel() // B
            L_e, n_batches = 0., 0
            for i in range(n_full):
                sel = idx[i * B:(i + 1) * B]
                batch = [d[sel] for d in data]
                if update_shape_constants is not None:
                    update_shape_constants(batch)
                loss, loss_for_gradient = infr_executor(*batch)
                loss_for_gradient.backward()
                if verbose:
                    print('\repoch {} Iteration {} loss: {}\t\t\t'.format(e + 1, i + 1, float(loss.detach())), end='')
                trainer.step(batch_size=B)
                L_e += float(loss.detach())
                n_batches += 1
            carry = idx[n_full * B:]
            if verbose and n_batches:
                print('epoch-loss: {} '.format(L_e / n_batches))
