"""Noise source for reparameterised sampling: the injection seam of
mxfusion/components/distributions/random_gen.py:21-98 (tests use a mock that replays a caller-supplied
buffer, util/testutils.py:58-93).  Drawing the N(0,1) noise itself is plumbing (torch's Philox on the device);
everything downstream of eps is HIP."""
import torch


class RandomGenerator(object):
    @staticmethod
    def sample_normal(loc=0, scale=1, shape=None, dtype=None, out=None, ctx=None, F=None):
        raise NotImplementedError


class TorchRandomGenerator(RandomGenerator):
    @staticmethod
    def sample_normal(loc=0, scale=1, shape=None, dtype=None, out=None, ctx=None, F=None):
        from ...common import config
        eps = torch.randn(tuple(shape), dtype=config.torch_dtype(dtype), device=ctx or config.get_default_device())
        if scale != 1:
            eps = eps * scale
        if loc != 0:
            eps = eps + loc
        return eps


MXNetRandomGenerator = TorchRandomGenerator   # source-compatible alias


class MockRandomGenerator(RandomGenerator):
    """Replays `samples` (flattened, cycled) -- util/testutils.py:58-93 MockMXNetRandomGenerator."""

    def __init__(self, samples):
        self._samples = samples.reshape(-1)
        self._pos = 0

    def sample_normal(self, loc=0, scale=1, shape=None, dtype=None, out=None, ctx=None, F=None):
        n = 1
        for s in shape:
            n *= int(s)
        idx = (torch.arange(n, device=self._samples.device) + self._pos) % self._samples.numel()
        self._pos = (self._pos + n) % self._samples.numel()
        return self._samples[idx].reshape(tuple(shape)).clone()


MockMXNetRandomGenerator = MockRandomGenerator
