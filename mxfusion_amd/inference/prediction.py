"""ModulePredictionAlgorithm (mxfusion/inference/prediction.py:24-85)."""
from .inference_alg import SamplingAlgorithm
from ..common.exceptions import InferenceError


class ModulePredictionAlgorithm(SamplingAlgorithm):
    def compute(self, F, variables):
        from ..components.distributions.distribution import Distribution
        from ..components.functions.function_evaluation import FunctionEvaluation
        from ..modules.module import Module
        outcomes = {}
        for f in self.model.ordered_factors:
            if isinstance(f, FunctionEvaluation):
                outcome = f.eval(F=F, variables=variables, always_return_tuple=True)
                for v, (_, ov) in zip(outcome, f.outputs):
                    variables[ov.uuid] = v
                    outcomes[ov.uuid] = v
            elif isinstance(f, Distribution):
                known = [v.uuid in variables for _, v in f.outputs]
                if all(known):
                    continue
                elif any(known):
                    raise InferenceError('Part of the outputs of the distribution %s has been observed!' % f.__class__.__name__)
                outcome = f.draw_samples(F=F, num_samples=self.num_samples, variables=variables, always_return_tuple=True)
                for v, (_, ov) in zip(outcome, f.outputs):
                    variables[ov.uuid] = v
                    outcomes[ov.uuid] = v
            elif isinstance(f, Module):
                outcome_uuid = [v.uuid for _, v in f.outputs]
                outcome = f.predict(F=F, variables=variables, targets=outcome_uuid, num_samples=self.num_samples)
                for v, uuid in zip(outcome, outcome_uuid):
                    variables[uuid] = v
                    outcomes[uuid] = v
        if self.target_variables:
            tv = [t.uuid if hasattr(t, 'uuid') else t for t in self.target_variables]
            return tuple(outcomes[u] for u in tv)
        return outcomes
