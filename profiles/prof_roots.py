import sys, ctypes as C
sys.path.insert(0,'wildcat-slam_amd/python')
import numpy as np
from wildcat_slam_amd import lib, synth
ctx=lib.Context(0)
pts,_=synth.g2_lattice(3906,m=32)
for i in range(3): s,ids=ctx.extract_surfels(pts)
out=(C.c_uint32*64)()
ctx.lib.wc_debug_status(ctx.h, out)
v=np.array(out[16:23],dtype=np.float64); print("roots timed", out[24])
names=["head->loads","staging","seq pass","flush+fence","moments","eigen","gates+stores"]
for n,c in zip(names,v): print(f"{n:14s} {c:10.0f} cycles/root  {c/2400:7.2f} us @2.4GHz")
print("sum us", v.sum()/2400)
