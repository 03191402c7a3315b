import os, sys, numpy as np, ctypes as C
sys.path.insert(0, os.getcwd())
from feabas_amd import _lib
lib, ctx = _lib.load(), _lib.ctx()
ident = np.zeros(128, dtype=np.uint8)
print('uid rc', lib.fb_comm_unique_id(ctx, _lib.ptr(ident)), lib.fb_last_error(ctx))
h = C.c_void_p()
rc = lib.fb_comm_create(ctx, _lib.ptr(ident), 0, 1, C.byref(h))
print('create rc', rc, lib.fb_last_error(ctx))
