"""Stiffness functions of materials (feabas/material.py:24-36, 128-131, 546-551): a material may scale its stiffness by a
function of the area stretch of each triangle -- the "wrinkle" material of configs/default_material_table.yaml:46-56 is soft
against expansion and stiff against compression.  The reference accepts any Python callable (material.py:128-131,
common.str_to_func): piecewise-linear tables -- what the one factory the reference ships, ``asymmetrical_elasticity``,
produces -- are evaluated on the device; any other callable is evaluated on the host on the area stretches of the
triangles and reaches the device as per-triangle multipliers (Mesh.assemble_into)."""
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
    StiffnessTable (device-evaluated), any other callable f(area stretch) -> multiplier (host-evaluated), None for no function.
    Like common.str_to_func (common.py:467-491): a callable, a 'lambda ...' string, or the dotted name of a plug-in; if calling
    it with the parameters yields a callable it was a factory and the product is the function.  '<lambda_bytes>' strings
    need dill, which is not in the image."""
    params = params or {}
    if factory is None:
        return None
    if isinstance(factory, StiffnessTable):
        return factory
    if callable(factory) and getattr(factory, '__name__', '') == 'asymmetrical_elasticity':
        return asymmetrical_elasticity(**params)
    if isinstance(factory, str) and factory.rsplit('.', 1)[-1] == 'asymmetrical_elasticity':
        return asymmetrical_elasticity(**params)
    func = factory
    if isinstance(factory, str):
        if factory.startswith('<lambda_bytes>'):
            raise NotImplementedError("stiffness function serialised with dill ('<lambda_bytes>...'): dill is not in the image")
        if factory.startswith('lambda'):
            func = eval(factory)                               # noqa: S307 -- the reference's own rule (common.py:478-479) for material tables
        else:
            import importlib
            mod, _, name = factory.rpartition('.')
            func = getattr(importlib.import_module(mod), name)
    if not callable(func):
        raise TypeError(f'stiffness function factory {factory!r} is not callable')
    try:                                                       # a factory? (common.py:484-490)
        produced = func(**params)
        if callable(produced):
            func = produced
    except Exception:                                          # noqa: BLE001 -- the reference's rule: not a factory, the function itself
        pass
    return func
