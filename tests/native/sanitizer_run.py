"""Driven by tests/test_native_sanitizers.py in a subprocess with libasan preloaded: runs the host-compiled kernel code
(relpose_core.h / relpose_rounds.h / guided_wave.h) built with -fsanitize=address,undefined over small and degenerate
inputs, every speculation width, single pairs and batches.  argv: the two sanitized shared objects."""
import ctypes as C, numpy as np, sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import test_relpose_core_host as rp
import test_guided_host as gh
host=C.CDLL(sys.argv[1])
P=rp._p
rng=np.random.default_rng(5)
for width in (1,3,16):
    host.host_set_max_width(width)
    for trial in range(12):
        b1,b2,_=rp._scene(rng,5,outliers=0.0,noise=1e-2 if trial%2 else 0.0)
        Es=np.zeros(90); host.host_essential_five_points(P(b1,C.c_double),P(b2,C.c_double),P(Es,C.c_double))
    z=np.zeros((5,3)); host.host_essential_five_points(P(z,C.c_double),P(z,C.c_double),P(np.zeros(90),C.c_double))
    for n,outl in ((5,0.0),(9,0.2),(30,0.3),(200,0.5),(4,0.0),(0,0.0)):
        b1,b2,_=rp._scene(rng,n,outliers=outl) if n else (np.zeros((0,3)),np.zeros((0,3)),None)
        model,lo,inl,it=np.zeros(12),np.zeros(12),np.zeros(max(n,1),np.int32),C.c_int(0)
        host.host_ransac_relative_pose(width,P(b1,C.c_double),P(b2,C.c_double),n,C.c_double(0.004),1000,C.c_double(0.99),1,10,P(model,C.c_double),P(lo,C.c_double),P(inl,C.c_int32),C.byref(it))
        R,t,models,info=np.zeros(9),np.zeros(3),np.zeros(24),np.zeros(2,np.int32); mask=np.zeros(max(n,1),np.uint8)
        host.host_robust_match_calibrated(P(b1,C.c_double),P(b2,C.c_double),n,C.c_double(0.004),1000,C.c_double(0.99),1,10,10,P(R,C.c_double),P(t,C.c_double),P(mask,C.c_uint8),P(models,C.c_double),P(info,C.c_int32))
    # a batch of pairs sharing the work lists
    sizes=[4,5,9,30,77,0,120]
    sc=[rp._scene(rng,n,outliers=0.3) if n else (np.zeros((0,3)),np.zeros((0,3)),None) for n in sizes]
    bb1=np.ascontiguousarray(np.concatenate([q[0] for q in sc])); bb2=np.ascontiguousarray(np.concatenate([q[1] for q in sc]))
    off=np.r_[0,np.cumsum(sizes)].astype(np.int64)
    scores,iters=np.zeros(len(sizes),np.int32),np.zeros(len(sizes),np.int32); models,mask=np.zeros((len(sizes),24)),np.zeros(len(bb1),np.uint8)
    host.host_rounds_ransac_batch(P(bb1,C.c_double),P(bb2,C.c_double),P(off,C.c_int64),len(sizes),C.c_double(0.004),1000,C.c_double(0.99),1,10,P(scores,C.c_int32),P(iters,C.c_int32),P(models,C.c_double),P(mask,C.c_uint8))
    for model,par in list(__import__('test_oracle_relpose')._BEARING_CAMERAS.items()):
        import oracle
        px=np.ascontiguousarray(rng.uniform(-0.4,0.4,(200,2))); p16=np.r_[np.asarray(par,float),np.zeros(16-len(par))]; out=np.zeros((200,3))
        host.host_pixel_bearings_generic(oracle.BEARING_MODELS[model],P(p16,C.c_double),P(px,C.c_double),200,P(out,C.c_double))
print("relpose harness under ASan/UBSan: clean")
g=C.CDLL(sys.argv[2])
for n in (1,2,60,130):
    d1,d2,b1,b2,R,o,perm=gh.guided_scene(rng,n)
    gh.host_match(g,d1,d2,b1=b1,b2=b2,R=R,t=o,threshold=0.01)
    m=(rng.random((len(d1),len(d2)))<0.2)
    gh.host_match(g,d1,d2,mask=m,ratio=0.9,symmetric=False)
print("guided harness under ASan/UBSan: clean")
