"""Factor: a node with named input and output Variables (mxfusion/components/factor.py:76-99: attribute
access by input/output name)."""
import uuid as _uuid


class Factor(object):
    def __init__(self, inputs, outputs, input_names, output_names):
        object.__setattr__(self, '_inputs', list(inputs) if inputs is not None else [])
        object.__setattr__(self, '_outputs', list(outputs) if outputs is not None else [])
        self.uuid = str(_uuid.uuid4()).replace('-', '_')
        self._input_names = list(input_names) if input_names is not None else [n for n, _ in self._inputs]
        self._output_names = list(output_names) if output_names is not None else [n for n, _ in self._outputs]
        self.graph = None
        for _, v in self._outputs:
            v.factor = self

    @property
    def inputs(self):
        return self._inputs

    @property
    def outputs(self):
        return self._outputs

    @property
    def input_names(self):
        return self._input_names

    @property
    def output_names(self):
        return self._output_names

    def __repr__(self):
        """factor.py:113-119: ClassName(input=variable, ...)."""
        ins = self.__dict__.get('_inputs') or []
        return type(self).__name__ + ('(' + ', '.join('%s=%s' % (n, v) for n, v in ins) + ')' if ins else '')

    def set_single_output(self, var):
        self._outputs = [(self._output_names[0], var)]
        var.factor = self

    def __getattr__(self, name):
        if name.startswith('_'):
            raise AttributeError(name)
        for n, v in self.__dict__.get('_inputs', []):
            if n == name:
                return v
        for n, v in self.__dict__.get('_outputs', []):
            if n == name:
                return v
        raise AttributeError("'%s' object has no attribute '%s'" % (type(self).__name__, name))
