"""Helpers of mxfusion/util/inference.py:23-87 (UUID normalisation, shape realisation)."""
from ..components.variables.variable import Variable


def variables_to_UUID(variables):
    return [(v.uuid if isinstance(v, Variable) else v) for v in variables]


def realize_shape(shape, constants):
    out = []
    for s in shape:
        if isinstance(s, Variable):
            if s.uuid in constants:
                out.append(int(constants[s.uuid]))
            elif s.isConstant:
                out.append(int(s.constant))
            else:
                raise ValueError('unresolved symbolic dimension %r' % (s,))
        else:
            out.append(int(s))
    return tuple(out)


class VariablesDict(dict):
    """Runtime {uuid: array} dict that also accepts Variable keys (the reference gets this from
    ModelComponent.__hash__/__eq__ hashing the UUID, model_component.py:53-57)."""

    @staticmethod
    def _k(key):
        return key.uuid if isinstance(key, Variable) else key

    def __getitem__(self, key):
        return dict.__getitem__(self, self._k(key))

    def __setitem__(self, key, value):
        dict.__setitem__(self, self._k(key), value)

    def __contains__(self, key):
        return dict.__contains__(self, self._k(key))

    def get(self, key, default=None):
        return dict.get(self, self._k(key), default)
