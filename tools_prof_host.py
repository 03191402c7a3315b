import cProfile, pstats, sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from feabas_amd import _lib
from feabas_amd.stitch_pipeline import StripBatchMatcher
lib, ctx = _lib.load(), _lib.ctx()
P,H,W=32,4096,510
s0=_lib.DeviceBuffer(P*H*W); s1=_lib.DeviceBuffer(P*H*W); sh=_lib.DeviceBuffer(P*8)
_lib.check(lib.fb_synth_strips_dev(ctx,P,0,H,W,2026,20,s0.ptr,s1.ptr,sh.ptr))
m=StripBatchMatcher(P,H,W)
m.match(s0.ptr,s1.ptr); m.match(s0.ptr,s1.ptr)
pr=cProfile.Profile(); pr.enable()
for _ in range(3): m.match(s0.ptr,s1.ptr)
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(25)
