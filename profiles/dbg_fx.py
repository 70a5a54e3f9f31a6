import sys, os
sys.path[:0]=["/root/repo/wildcat-slam_amd/python","/root/repo/oracle","/root/repo/tests"]
import numpy as np, pyoracle, helpers
from wildcat_slam_amd import lib, synth
ctx=lib.Context(0)
which=sys.argv[1]
if which=="room": pts=synth.g1_room(int(sys.argv[2]))
elif which=="g2": pts,_=synth.g2_lattice(int(sys.argv[2]),m=32)
elif which=="g2x2":
    a,_=synth.g2_lattice(int(sys.argv[2]),m=32,t_start=0.0,duration=0.1); b,_=synth.g2_lattice(int(sys.argv[2]),m=32,t_start=0.3,duration=0.1)
    b["x"]+=np.float32(0.001); pts=synth.concat_points(a,b)
params=pyoracle.default_params()
if len(sys.argv)>3: params.max_layer=int(sys.argv[3]); ctx.set_params(params)
s_ref,i_ref,st=pyoracle.extract_surfels(pts,params)
print("oracle", len(s_ref), list(st.nodes_tested), list(st.nodes_plane), flush=True)
s,i=ctx.extract_surfels(pts)
print("gpu", len(s), ctx.extract_path_info(), flush=True)
ta=set(helpers.id_tuples(i)); tb=set(helpers.id_tuples(i_ref))
miss=sorted(tb-ta); extra=sorted(ta-tb)
print("missing",len(miss),"extra",len(extra))
import collections
print("missing by (layer, ord):", collections.Counter(((m[3]&3),(m[3]>>8)) for m in miss).most_common(12))
print("extra by (layer, ord):", collections.Counter(((m[3]&3),(m[3]>>8)) for m in extra).most_common(12))
if not miss and not extra:
    print(helpers.check_surfels(s,i,s_ref,i_ref,tol=1e-6,t_tol=1e-4))
