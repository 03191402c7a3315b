"""ctypes binding of libfeabas_hip.so (include/feabas_hip.h).

There is no CPU fallback: if the library is missing or no MI355X is visible,
every compute entry point raises.  Build with ``python -c "import
__graft_entry__ as g; g.build()"`` or ``make -C feabas_amd/csrc``.
"""
import ctypes as C
import os
import threading
import re

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# FEABAS_HIP_LIB: an A/B build of the library (tools/build_variant.sh); never needed in production
LIB_PATH = os.path.abspath(os.environ['FEABAS_HIP_LIB']) if os.environ.get('FEABAS_HIP_LIB') else os.path.join(_HERE, 'libfeabas_hip.so')
HEADER_PATH = os.path.join(os.path.dirname(_HERE), 'include', 'feabas_hip.h')


class FeabasHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f'libfeabas_hip error {code}: {msg}')
        self.code = code


FB_ERR_NOCONV = -5
FB_ERR_BREAKDOWN = -6
FB_ERR_COMM = -7

_lib = None
_ctx = None
_ctx_device = None


def cpu_budget():
    """CPUs this process may really use: the smaller of its affinity mask and its cgroup quota (a GPU box shows 256 logical
    CPUs to a container whose quota is 16: pools sized by os.cpu_count() then spin their way through the quota and every
    host thread -- the ones that feed the device included -- is throttled for tens of milliseconds at a time)"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ('/sys/fs/cgroup/cpu.max',):
        try:
            quota, period = open(path).read().split()[:2]
            if quota != 'max':
                n = min(n, max(1, int(int(quota) / int(period))))
        except (OSError, ValueError):
            pass
    try:
        q = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read()); per = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
        if q > 0:
            n = min(n, max(1, q // per))
    except (OSError, ValueError):
        pass
    # the ranks of one node share the mask and the quota (one process per GPU under torch.distributed.run)
    try:
        local_world = int(os.environ.get('LOCAL_WORLD_SIZE', '1'))
    except ValueError:
        local_world = 1
    return max(1, n // max(1, local_world))


_pool_limiter = None


def limit_host_pools(limit=None):
    """Caps the BLAS / OpenMP pools of the process at a quarter of cpu_budget() (at least 1, at most 8): numpy's OpenBLAS
    starts one spinning thread per logical CPU it sees, and on the host side of the two paths nothing larger than a 3 x 3
    system goes through BLAS.  Measured on a 16-CPU-quota / 256-CPU box: SLM.optimize_linear of a 500 k-DoF section 50-80 ms
    with the default pool, 11-12 ms with it capped (tools/prof_align_section.py).
    Not applied at import (importing the accelerator must not slow the host code of the embedding application): the first
    device context applies it, ``restore_host_pools()`` undoes it, FEABAS_HIP_KEEP_POOLS=1 leaves the pools alone for good.
    No environment variable is written: child processes and libraries loaded later keep their own defaults."""
    global _pool_limiter
    if os.environ.get('FEABAS_HIP_KEEP_POOLS') or _pool_limiter is not None:
        return None
    lim = max(1, min(8, cpu_budget() // 4)) if limit is None else int(limit)
    try:
        from threadpoolctl import threadpool_limits
        _pool_limiter = threadpool_limits(limits=lim)
    except Exception:           # threadpoolctl absent: nothing to cap
        return None
    return lim


def restore_host_pools():
    """give the BLAS / OpenMP pools their sizes of before the first device context back"""
    global _pool_limiter
    if _pool_limiter is not None:
        try:
            _pool_limiter.restore_original_limits()
        except Exception:
            pass
        _pool_limiter = None


c_p = C.c_void_p
c_i = C.c_int
c_i64 = C.c_int64
c_d = C.c_double
c_sz = C.c_size_t

_PROTOS = {
    'fb_create': (c_p, [c_i]),
    'fb_destroy': (None, [c_p]),
    'fb_last_error': (C.c_char_p, [c_p]),
    'fb_sync': (c_i, [c_p]),
    'fb_stream': (c_p, [c_p]),
    'fb_device_info': (c_i, [c_p, C.c_char_p, c_i, C.POINTER(c_i), C.POINTER(c_sz)]),
    'fb_version': (C.c_char_p, []),
    'fb_malloc': (c_i, [c_p, c_sz, C.POINTER(c_p)]),
    'fb_free': (c_i, [c_p, c_p]),
    'fb_memcpy_h2d': (c_i, [c_p, c_p, c_p, c_sz]),
    'fb_host_alloc': (c_i, [c_p, C.c_size_t, C.POINTER(c_p)]),
    'fb_host_free': (c_i, [c_p, c_p]),
    'fb_host_pack2d': (c_i, [c_p, c_p, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_i]),
    'fb_memcpy_d2h': (c_i, [c_p, c_p, c_p, c_sz]),
    'fb_memcpy_d2d': (c_i, [c_p, c_p, c_p, C.c_size_t]),
    'fb_memcpy2d_d2d': (c_i, [c_p, c_p, c_sz, c_p, c_sz, c_sz, c_sz]),
    'fb_memset': (c_i, [c_p, c_p, c_i, c_sz]),
    'fb_timer_start': (c_i, [c_p]),
    'fb_timer_stop': (c_i, [c_p, C.POINTER(C.c_float)]),
    'fb_prof_enable': (c_i, [c_p, c_i]),
    'fb_prof_reset': (c_i, [c_p]),
    'fb_prof_count': (c_i, [c_p]),
    'fb_prof_get': (c_i, [c_p, c_i, C.c_char_p, c_i, C.POINTER(c_i), C.POINTER(c_d), C.POINTER(c_d)]),
    'fb_next_fast_len': (c_i, [c_i]),
    'fb_ncc_batch': (c_i, [c_p, c_p, c_p] + [c_i] * 9 + [c_p, c_p, c_p]),
    'fb_ncc_batch_dev': (c_i, [c_p, c_p, c_p] + [c_i] * 9 + [c_p, c_p, c_p]),
    'fb_ncc_batch_normalized': (c_i, [c_p, c_p, c_p] + [c_i] * 6 + [c_p, c_p] + [c_i] * 3 + [c_p, c_p, c_p]),
    'fb_ncc_blocks_dev': (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p]),
    'fb_ncc_blocks_affine_dev': (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p]),
    'fb_ncc_last_surfaces': (c_i, [c_p, c_p, c_p, C.POINTER(c_i), C.POINTER(c_i)]),
    'fb_dog': (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_d, c_p, c_i, c_p]),
    'fb_dog_dev': (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_d, c_p, c_i, c_p]),
    'fb_area_resize_size': (c_i, [c_i, c_d]),
    'fb_area_resize_dev': (c_i, [c_p, c_p, c_i, c_i, c_i, c_d, c_d, c_p]),
    'fb_area_resize': (c_i, [c_p, c_p, c_i, c_i, c_i, c_d, c_d, c_p]),
    'fb_area_downsample2': (c_i, [c_p, c_p, c_i, c_i, c_i, c_p]),
    'fb_area_downsample2_dev': (c_i, [c_p, c_p, c_i, c_i, c_i, c_p]),
    'fb_area_downsample2_sizes_dev': (c_i, [c_p, c_p, c_i, c_i, c_i, c_p, c_p]),
    'fb_dog_sizes_dev': (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_d, c_i, c_p]),
    'fb_dog_masks_dev': (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_d, c_p, c_i, c_p]),
    'fb_dog_down2_dev': (c_i, [c_p, c_p, c_i, c_i, c_i, c_d, c_i, c_p]),
    'fb_dog_pair_dev': (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_d, c_i, c_p]),
    'fb_dog_down2_pair_dev': (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_d, c_i, c_p]),
    'fb_mask_range_dev': (c_i, [c_p, c_p, c_sz, C.c_float, C.c_float, c_p]),
    'fb_count_nonzero_dev': (c_i, [c_p, c_p, c_sz, C.POINTER(c_i64)]),
    'fb_divide_bbox': (c_i, [c_p, c_p, c_p, c_p, c_d, c_i, c_p, c_p, c_p, c_i, c_p, c_i]),
    'fb_mesh_block_affines': (c_i, [c_p, c_i, c_p, c_p, c_p, c_i, c_p, c_i, c_i, c_i, c_p, c_p, c_d, c_p, c_p]),
    'fb_mesh_block_uncovered_dev': (c_i, [c_p, c_p, c_p, c_i, c_p, c_i, c_i, c_i, c_p, c_p, c_p]),
    'fb_tri_edge_ratio': (c_i, [c_p, c_i, c_p, c_p, c_i, c_p, c_p]),
    'fb_signed_area': (c_i, [c_p, c_i, c_p, c_i, c_p, c_p]),
    'fb_mesh_block_uncovered': (c_i, [c_p, c_i, c_p, c_p, c_i, c_p, c_i, c_i, c_i, c_p, c_p, c_p]),
    'fb_mesh_locate_dev': (c_i, [c_p, c_i, c_p, c_p, c_i, c_p, c_p]),
    'fb_mesh_candidates_dev': (c_i, [c_p, c_i, c_p, c_p, c_i, c_p, c_i, c_i, c_i, c_p, c_p]),
    'fb_mesh_render_blocks_dev': (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_i, c_p, c_i, c_i, c_p, c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_p]),
    'fb_remap_dev': (c_i, [c_p, c_p, c_i, c_i, c_i, c_p, c_i, c_i, c_p, c_p, c_p, c_p, c_p]),
    'fb_synth_strips_dev': (c_i, [c_p, c_i, c_i, c_i, c_i, C.c_uint32, c_i, c_i, C.c_float, c_p, c_p, c_p]),
    'fb_sys_create': (c_i, [c_p, c_i64, C.POINTER(c_p)]),
    'fb_sys_destroy': (None, [c_p, c_p]),
    'fb_sys_add_mesh': (c_i, [c_p, c_p, c_i64, c_p, c_i, c_i, C.POINTER(c_i)]),
    'fb_sys_set_links': (c_i, [c_p, c_p, c_i64, c_p]),
    'fb_sys_finalize': (c_i, [c_p, c_p, C.POINTER(c_i64)]),
    'fb_schedule_create': (c_p, [c_p, c_i, c_i, c_i, c_i, c_i]),
    'fb_schedule_destroy': (None, [c_p]),
    'fb_schedule_round': (c_i, [c_p, C.POINTER(c_d), C.POINTER(c_i), C.POINTER(c_i)]),
    'fb_schedule_advance': (c_i, [c_p, c_d, c_d, C.POINTER(c_i)]),
    'fb_sys_pattern': (c_i, [c_p, c_p, c_p, c_p]),
    'fb_sys_info': (c_i, [c_p, c_p, C.POINTER(c_i64), C.POINTER(c_i64), C.POINTER(c_i64)]),
    'fb_sys_assemble_mesh': (c_i, [c_p, c_p, c_i, c_p, c_p, c_p, c_d, c_d]),
    'fb_sys_assemble_mesh_add': (c_i, [c_p, c_p, c_i, c_p, c_p, c_p, c_d, c_d]),
    'fb_sys_assemble_mesh_materials': (c_i, [c_p, c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_d]),
    'fb_sys_assemble_mesh_stretch': (c_i, [c_p, c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_p, c_p, c_p, c_d]),
    'fb_sys_assemble_links': (c_i, [c_p, c_p, c_p, c_p, c_p]),
    'fb_sys_update_links': (c_i, [c_p, c_p, c_i64, c_p]),
    'fb_sys_form_groups': (c_i, [c_p, c_p, c_i, c_d, c_d, c_p]),
    'fb_sys_solve_groups': (c_i, [c_p, c_p, c_i, c_p, c_d, c_d, c_i, c_i, c_p, c_p]),
    'fb_sys_group_energy': (c_i, [c_p, c_p, c_i, c_p, c_p]),
    'fb_pairs_relax': (c_i, [c_p, c_p, c_i, c_i, c_i, c_p, c_p, c_i64, c_p, c_p, c_p, c_p, c_p, c_d, c_i, c_d, c_d, c_d, c_p, c_p, c_p, c_p]),
    'fb_pairs_relax_bary': (c_i, [c_p, c_p, c_i, c_i64, c_p, c_p, c_p, c_p, c_d, c_i, c_d, c_p, c_d, c_d, c_p, c_p, c_p, c_p]),
    'fb_pairs_strain_bary': (c_i, [c_p, c_p, c_i, c_i64, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_d, c_p, c_d, c_p, c_p, c_p]),
    'fb_deformed_block_affines': (c_i, [c_p, c_i, c_i, c_i, c_p, c_p, c_i, c_p, c_i, c_p, c_d, c_p, c_p, c_p, c_p]),
    'fb_deformed_exact_field': (c_i, [c_p, c_i, c_i, c_i, c_p, c_p, c_i, c_p, c_i, c_p, c_p, c_i, c_i, c_p, c_p, c_p]),
    'fb_deformed_locate': (c_i, [c_p, c_i, c_i, c_i, c_p, c_p, c_i, c_p, c_i64, c_p, c_p, c_p, c_p]),
    'fb_hbm_probe': (c_i, [c_p, c_i, c_i, c_p]),
    'fb_strip_matcher_create': (c_i, [c_p, c_i, c_i, c_i, c_p, c_p]),
    'fb_strip_matcher_create_ragged': (c_i, [c_p, c_i, c_i, c_i, c_p, c_p, c_p]),
    'fb_strip_matcher_destroy': (None, [c_p, c_p]),
    'fb_strip_matcher_info': (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    'fb_match_strips': (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    'fb_match_strips_table': (c_i, [c_p, c_p, c_p, c_p, c_p, c_p]),
    'fb_match_strips_deformed': (c_i, [c_p, c_p, c_p, c_p, c_p]),
    'fb_strip_matcher_set_extras': (c_i, [c_p, c_p, c_p, c_p, c_i]),
    'fb_match_strips_photometric': (c_i, [c_p, c_p, c_p, c_p]),
    'fb_match_strips_field': (c_i, [c_p, c_p, c_p, c_p]),
    'fb_link_terms': (c_i, [c_p, c_i64, c_p, c_i64, c_p, c_p, c_p, c_i64, c_p, c_i64, c_p, c_p, c_p, c_i64, c_d, c_d, c_p, c_p, c_p]),
    'fb_pairs_strain': (c_i, [c_p, c_p, c_i, c_i, c_i, c_p, c_p, c_i64, c_p, c_p, c_p, c_p, c_p, c_d, c_d, c_i, c_d, c_p, c_p, c_p]),
    'fb_sys_lambda': (c_i, [c_p, c_p, c_d, c_d, C.POINTER(c_d), C.POINTER(c_d)]),
    'fb_sys_form': (c_i, [c_p, c_p, c_d, c_d]),
    'fb_sys_solve': (c_i, [c_p, c_p, c_p, c_i, c_d, c_d, c_i, c_i, C.POINTER(c_i), C.POINTER(c_d)]),
    'fb_sys_solve_fixed': (c_i, [c_p, c_p, c_i, C.POINTER(c_d)]),
    'fb_sys_get': (c_i, [c_p, c_p, c_i, c_p]),
    'fb_pcg': (c_i, [c_p, c_i64, c_p, c_p, c_p, c_p, c_p, c_i, c_d, c_d, c_i, c_i, c_i, C.POINTER(c_i), C.POINTER(c_d)]),
    'fb_csr_upload': (c_i, [c_p, c_i64, c_p, c_p, c_p, c_i, C.POINTER(c_p)]),
    'fb_csr_destroy': (None, [c_p, c_p]),
    'fb_csr_info': (c_i, [c_p, c_p] + [C.POINTER(c_i64)] * 4),
    'fb_spmv': (c_i, [c_p, c_p, c_p, c_p]),
    'fb_spmv_dev': (c_i, [c_p, c_p, c_p, c_p]),
    'fb_pcg_csr': (c_i, [c_p, c_p, c_p, c_p, c_i, c_d, c_d, c_i, c_i, C.POINTER(c_i), C.POINTER(c_d)]),
    'fb_pcg_fixed_iters': (c_i, [c_p, c_p, c_p, c_i, C.POINTER(c_d)]),
    'fb_cgcg_update_dev': (c_i, [c_p, c_i64, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    'fb_cgcg_dots_dev': (c_i, [c_p, c_i64, c_p, c_p, c_p, c_p, c_p]),
    'fb_cgcg_scalars_dev': (c_i, [c_p, c_p, c_p, c_i]),
    'fb_gather_f64_dev': (c_i, [c_p, c_i64, c_p, c_p, c_p]),
    'fb_cgcg_solve_dev': (c_i, [c_p, c_p, c_p, c_i64, c_i64, c_p, c_p, c_p, c_i, c_p, c_p, c_p, c_i, c_p, c_p, c_d, c_i, c_i, C.POINTER(c_i), C.POINTER(c_d), C.POINTER(c_d)]),
    'fb_comm_unique_id': (c_i, [c_p, c_p]),
    'fb_comm_create': (c_i, [c_p, c_p, c_i, c_i, C.POINTER(c_p)]),
    'fb_comm_destroy': (None, [c_p, c_p]),
    'fb_comm_info': (c_i, [c_p, c_p, C.POINTER(c_i), C.POINTER(c_i)]),
    'fb_gatherv_dev': (c_i, [c_p, c_p, c_p, c_p, c_p, c_i]),
    'fb_allgather_dev': (c_i, [c_p, c_p, c_p, c_p, c_sz]),
    'fb_allreduce_f64_dev': (c_i, [c_p, c_p, c_p, c_p, c_sz, c_i]),
    'fb_sendrecv_dev': (c_i, [c_p, c_p, c_i, c_p, c_p, c_p, c_i, c_p, c_p, c_p]),
}


# test hooks: only in libfeabas_hip_test.so (include/feabas_hip_test.h), loaded by tests and fuzzers through load_test()
_TEST_PROTOS = {
    'fb_debug_rigid_fits': (c_i, [c_i, c_i64, c_p, c_p, c_p, c_p, c_p, c_p]),
    'fb_debug_auto_spacings': (c_i, [c_i, c_i, c_p, c_i, c_p]),
    'fb_debug_grid_counts': (c_i, [c_i, c_i, c_d, c_i, c_p, c_p]),
    'fb_debug_mg_coarsen': (c_i, [c_i, c_i, c_p, c_p, c_p, c_p, c_d, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_p]),
    'fb_debug_fft1d': (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i]),
}


def declared_symbols(header=None):
    """Every function name include/feabas_hip.h (or another header of include/) declares."""
    with open(HEADER_PATH if header is None else os.path.join(os.path.dirname(HEADER_PATH), header)) as f:
        txt = f.read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(fb_[a-z0-9_]+)\s*\(', txt)))


class StripOpts(C.Structure):
    """fb_strip_opts (include/feabas_hip.h)"""
    _fields_ = [('sigma', C.c_double), ('coarse_downsample2', C.c_int), ('conf_thresh', C.c_double), ('min_num_blocks', C.c_int),
                ('conf_mode', C.c_int), ('residue_len', C.c_double), ('residue_mode', C.c_int), ('stiffness_lambda', C.c_double),
                ('relax_tol', C.c_double), ('compute_strain', C.c_int), ('nspacings', C.c_int), ('spacings', C.c_void_p)]


def load():
    """Load the shared library (no GPU needed for this step)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f'{LIB_PATH} not found: build it with __graft_entry__.build() '
                          '(feabas_amd has no CPU fallback)')
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _PROTOS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


_test_lib = None
TEST_LIB_PATH = os.path.abspath(os.environ['FEABAS_HIP_TEST_LIB']) if os.environ.get('FEABAS_HIP_TEST_LIB') else os.path.join(_HERE, 'libfeabas_hip_test.so')      # (env: the sanitizer build of tools/asan_host.sh)


def load_test():
    """The test build of the library (the product objects plus the hooks of include/feabas_hip_test.h): a second, independent
    instance -- a hook that takes a context wants one made by THIS handle's fb_create.  Never loaded by the product."""
    global _test_lib
    if _test_lib is None:
        if not os.path.exists(TEST_LIB_PATH):
            raise ImportError(f'{TEST_LIB_PATH} not found: build it with __graft_entry__.build()')
        lib = C.CDLL(TEST_LIB_PATH)
        for name, (res, args) in list(_PROTOS.items()) + list(_TEST_PROTOS.items()):
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _test_lib = lib
    return _test_lib


_tls = threading.local()


def gpu_available():
    """whether this process can have a context (a GPU is visible and the library loads)"""
    try:
        ctx()
        return True
    except Exception:
        return False


def new_context(device=None):
    """An additional context on the process's device: its own HIP stream, scratch arena and event profile.  Host
    threads that drive the device concurrently use one each (``use_context``) so that a thread's synchronisation waits
    only for its own work."""
    lib = load()
    if device is None:
        device = _ctx_device if _ctx_device is not None else int(os.environ.get('FEABAS_HIP_DEVICE', os.environ.get('LOCAL_RANK', '0')))
    h = lib.fb_create(device)
    if not h:
        raise RuntimeError(f'fb_create({device}) failed')
    return h


def use_context(h):
    """Make `h` the calling thread's current context (None: back to the process context)."""
    _tls.ctx = h


class using:
    """``with using(h):`` -- run the body with `h` as the calling thread's current context"""

    def __init__(self, h):
        self.h = h

    def __enter__(self):
        self.prev = getattr(_tls, 'ctx', None)
        _tls.ctx = self.h
        return self.h

    def __exit__(self, *exc):
        _tls.ctx = self.prev
        return False


_destroy_hooks = []


def on_context_destroy(fn):
    """`fn(h)` is called just before a context goes away (``destroy_context``, or the process context being replaced by one on
    another device), with `h` still alive: modules that keep per-context objects (matchers, pools) let go of them there."""
    if fn not in _destroy_hooks:
        _destroy_hooks.append(fn)


def _run_destroy_hooks(h):
    for fn in list(_destroy_hooks):
        try:
            fn(h)
        except Exception:                                  # noqa: BLE001 -- a cache that cannot be emptied must not keep the context alive
            import traceback
            traceback.print_exc()


def destroy_context(h):
    """Destroy a context made by ``new_context`` (its stream, arena and every buffer it still owns)."""
    if h is not None and h != _ctx:
        _run_destroy_hooks(h)
        load().fb_destroy(h)


def ctx(device=None):
    """The current context: the calling thread's (``use_context``) or the per-process one (one GPU per process;
    LOCAL_RANK picks the device)."""
    global _ctx, _ctx_device
    h = getattr(_tls, 'ctx', None)
    if h is not None and device is None:
        return h
    lib = load()
    if device is None:
        device = int(os.environ.get('FEABAS_HIP_DEVICE', os.environ.get('LOCAL_RANK', '0')))
    if _ctx is not None and _ctx_device == device:
        return _ctx
    if _ctx is not None:
        _run_destroy_hooks(_ctx)
        lib.fb_destroy(_ctx)
        _ctx = None
    h = lib.fb_create(device)
    if not h:
        raise RuntimeError(f'fb_create({device}) failed: no usable MI355X/HIP device '
                           '(feabas_amd has no CPU fallback)')
    limit_host_pools()
    _ctx = h
    _ctx_device = device
    return _ctx


def check(rc, allow=(), h=None):
    """raise on a non-zero status; h = the context the call was made on (default: the calling thread's current one)"""
    if rc != 0 and rc not in allow:
        msg = load().fb_last_error(h)                      # the message belongs to the calling thread, whatever the context (host-only entries have none)
        raise FeabasHipError(rc, msg.decode() if msg else '?')
    return rc


def ptr(a):
    """Raw pointer of a C-contiguous numpy array (or None)."""
    if a is None:
        return None
    assert a.flags['C_CONTIGUOUS']
    return a.ctypes.data_as(c_p)


def as_c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


class DeviceBuffer:
    """A context-owned device allocation (plumbing for resident pipelines)."""

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        p = c_p()
        self.ctx = ctx()                                  # the owning context (a host thread may have its own)
        check(load().fb_malloc(self.ctx, self.nbytes, C.byref(p)))
        self.ptr = p

    @classmethod
    def from_array(cls, a):
        a = np.ascontiguousarray(a)
        buf = cls(a.nbytes)
        check(load().fb_memcpy_h2d(ctx(), buf.ptr, ptr(a), a.nbytes))
        return buf

    def count_nonzero(self, nbytes=None):
        """non-zero bytes among the first nbytes of the buffer, counted on the device (fb_count_nonzero_dev)"""
        n = c_i64()
        check(load().fb_count_nonzero_dev(ctx(), self.ptr, self.nbytes if nbytes is None else int(nbytes), C.byref(n)))
        return n.value

    def to_array(self, shape, dtype):
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes
        # on the CALLING thread's context (stream): a buffer is read by the thread that queued the kernels filling it --
        # matchers are built on one thread and driven from another, each with its own context (bench.py, matcher.py)
        check(load().fb_memcpy_d2h(ctx(), ptr(out), self.ptr, out.nbytes))
        return out

    def offset(self, nbytes):
        return c_p(self.ptr.value + int(nbytes))

    def free(self):
        if self.ptr is not None and _ctx is not None:
            load().fb_free(self.ctx, self.ptr)
        self.ptr = None


class PinnedBuffer:
    """page-locked host staging memory (fb_host_alloc) viewed as a numpy array"""

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        p = c_p()
        self.ctx = ctx()
        check(load().fb_host_alloc(self.ctx, self.nbytes, C.byref(p)))
        self.ptr = p
        self._raw = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_ubyte)), shape=(self.nbytes,))

    def array(self, shape, dtype, offset=0):
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        assert offset + n <= self.nbytes
        return self._raw[offset:offset + n].view(dtype).reshape(shape)

    def free(self):
        if self.ptr is not None and _ctx is not None:
            self._raw = None
            load().fb_host_free(self.ctx, self.ptr)
        self.ptr = None


def prof_snapshot(h=None):
    """{kernel name: (launches, total ms, total algorithmic bytes)} from the event profile of context h (default: current)."""
    lib = load()
    out = {}
    h = ctx() if h is None else h
    n = lib.fb_prof_count(h)
    name = C.create_string_buffer(128)
    launches = c_i()
    ms = c_d()
    nbytes = c_d()
    for i in range(n):
        check(lib.fb_prof_get(h, i, name, 128, C.byref(launches), C.byref(ms), C.byref(nbytes)))
        out[name.value.decode()] = (launches.value, ms.value, nbytes.value)
    return out
