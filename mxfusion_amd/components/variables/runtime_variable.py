"""The sample-axis convention (mxfusion/components/variables/runtime_variable.py:20-118).  Unlike the
reference, broadcasting over S is by stride 0 (`expand`), never materialised: the HIP kernels take a sample
stride of 0."""
import torch


def add_sample_dimension(F, array):
    return array.unsqueeze(0)


def add_sample_dimension_to_arrays(F, arrays, out=None):
    processed = {k: (v.unsqueeze(0) if isinstance(v, torch.Tensor) else v) for k, v in arrays.items()}
    if out is not None:
        out.update(processed)
    return processed


def expectation(F, array):
    return array.mean(dim=0)


def array_has_samples(F, array):
    return array.shape[0] > 1


def get_num_samples(F, array):
    return array.shape[0]


def as_samples(F, array, num_samples):
    if array_has_samples(F, array):
        return array
    return array.expand((num_samples,) + tuple(array.shape[1:]))


def arrays_as_samples(F, arrays):
    num = [max(get_num_samples(F, v) for v in a.values()) if isinstance(a, dict) else get_num_samples(F, a) for a in arrays]
    mx = max(num)
    if mx > 1:
        return [{k: as_samples(F, v, mx) for k, v in a.items()} if isinstance(a, dict) else as_samples(F, a, mx) for a in arrays]
    return arrays
