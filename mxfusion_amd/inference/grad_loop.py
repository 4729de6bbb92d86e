"""GradLoop interface (mxfusion/inference/grad_loop.py:19-45)."""


class GradLoop(object):
    def run(self, infr_executor, data, param_dict, ctx, optimizer='adam', learning_rate=1e-3, max_iter=2000, verbose=False):
        raise NotImplementedError
