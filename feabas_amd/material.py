"""Stiffness functions of materials (feabas/material.py:24-36, 128-131, 546-551): a material may scale its stiffness by a
function of the area stretch of each triangle -- the "wrinkle" material of configs/default_material_table.yaml:46-56 is soft
against expansion and stiff against compression.  The reference accepts any Python callable; the device evaluates
piecewise-linear tables, which is what the one factory the reference ships (``asymmetrical_elasticity``) produces."""
import numpy as np


class StiffnessTable:
    """y(x) through the knots (strain, stiffness), linear in between, constant beyond the ends -- scipy's
    interp1d(kind='linear', bounds_error=False, fill_value=(stiffness[0], stiffness[-1])) of material.py:549."""

    def __init__(self, strain, stiffness):
        self.strain = np.ascontiguousarray(strain, dtype=np.float64).ravel()
        self.stiffness = np.ascontiguousarray(stiffness, dtype=np.float64).ravel()
        if self.strain.size != self.stiffness.size or self.strain.size < 2 or np.any(np.diff(self.strain) <= 0):
            raise ValueError('a stiffness function needs >= 2 knots with ascending strain')

    def __call__(self, x):
        xs, ys = self.strain, self.stiffness
        x = np.asarray(x, dtype=np.float64)
        hi = np.clip(np.searchsorted(xs, x), 1, xs.size - 1)
        lo = hi - 1
        y = (ys[hi] - ys[lo]) / (xs[hi] - xs[lo]) * (x - xs[lo]) + ys[lo]
        return np.where(x < xs[0], ys[0], np.where(x > xs[-1], ys[-1], y))

    def __eq__(self, other):
        return isinstance(other, StiffnessTable) and np.array_equal(self.strain, other.strain) and np.array_equal(self.stiffness, other.stiffness)

    def __hash__(self):
        return hash((self.strain.tobytes(), self.stiffness.tobytes()))


def asymmetrical_elasticity(**params):
    """material.py:546-551; strain = 1: no change, strain = 0: flip."""
    return StiffnessTable(params.get('strain', [0, 0.75, 1, 1.01]), params.get('stiffness', [1.5, 1, 0.5, 0]))


def stiffness_func_from_spec(factory, params=None):
    """the `stiffness_func_factory` / `stiffness_func_params` entries of a material table (material.py:60-62, 128-131) -> a
    StiffnessTable, None for no function.  Factories other than asymmetrical_elasticity (plug-ins, lambdas) cannot run on the
    device and are refused."""
    if factory is None:
        return None
    if isinstance(factory, StiffnessTable):
        return factory
    if callable(factory) and getattr(factory, '__name__', '') == 'asymmetrical_elasticity':
        return asymmetrical_elasticity(**(params or {}))
    if isinstance(factory, str) and factory.rsplit('.', 1)[-1] == 'asymmetrical_elasticity':
        return asymmetrical_elasticity(**(params or {}))
    raise NotImplementedError(f'stiffness function factory {factory!r}: only piecewise-linear tables (asymmetrical_elasticity) run on the device')
