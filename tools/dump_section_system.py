"""the two linear systems a section pair solves (tools/bench_section_matcher.py workload), saved as scipy CSR + right-hand
side under gpurun_out/: material for preconditioner experiments on the host"""
import os, sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
import bench_section_matcher as bsm                          # noqa: E402
from feabas_amd import optimizer, _lib
from feabas_amd.mesh import bsr_download
from scipy import sparse
os.makedirs('gpurun_out/sysdump', exist_ok=True)
count = [0]
orig = optimizer.SLM.optimize_linear


def patched(self, **kw):
    out = orig(self, **kw)
    if count[0] < 2 and self._sys is not None:
        A = bsr_download(self._sys, 4, self._nv, self._nnzb).tocsr()
        b = np.empty(2 * self._nv)
        _lib.check(_lib.load().fb_sys_get(_lib.ctx(), self._sys, 5, _lib.ptr(b)))
        xy = np.concatenate([m.vertices(m._current_gear if hasattr(m, '_current_gear') else 1) for m in self.meshes if not m.locked])
        sparse.save_npz(f'gpurun_out/sysdump/A{count[0]}.npz', A)
        np.savez(f'gpurun_out/sysdump/b{count[0]}.npz', b=b, xy=xy, nv=[m.num_vertices for m in self.meshes], tol=kw.get('tol', 1e-7), iters=self.last_solve['iters'])
        print('dumped system', count[0], A.shape, 'iters', self.last_solve['iters'], 'tol', kw.get('tol'))
        count[0] += 1
    return out


optimizer.SLM.optimize_linear = patched
sys.argv = ['x']
bsm.main()
