import sys, ctypes as C, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import feabas_amd as fb
from feabas_amd import _lib
from oracle import fem_ref
import importlib.util
spec = importlib.util.spec_from_file_location('tf', 'tests/test_gpu_fem.py'); tf = importlib.util.module_from_spec(spec); spec.loader.exec_module(tf)
lib, ctx = _lib.load(), _lib.ctx()
for n, nl in ((40, 300), (90, 800)):
    for pre in (1, 2):
        rng = np.random.default_rng(3)
        prod, lp, _, _ = tf._random_system(fb, rng, n, n * 3 // 4, nl, two_free=True)
        slm = fb.optimizer.SLM(prod, lp)
        slm._assemble(0, 1, 1)
        sl, cl = slm.relative_lambda_trace(1.0, -1.0)
        _lib.check(lib.fb_sys_form(ctx, slm._sys, sl, cl))
        dd = np.zeros(2 * slm._nv); it, rr = C.c_int(), C.c_double()
        rc = lib.fb_sys_solve(ctx, slm._sys, _lib.ptr(dd), 0, 1e-8, 0.0, 5000, pre, C.byref(it), C.byref(rr))
        print(n, nl, 'pre', pre, 'rc', rc, 'iters', it.value, 'relres', rr.value, flush=True)
