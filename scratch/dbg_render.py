import sys, numpy as np
sys.path.insert(0, '.')
from phosphorus_mk2_amd import scenes, xpu
from oracle import oracle as orc
sc = scenes.cornell(64, 64)
for spp, depth in [(1,1),(1,2),(1,3),(1,9),(4,9)]:
    film, st = xpu.render(sc, spp=spp, pps=1, depth=depth, seed=1)
    ref, ost = orc.Oracle(sc, spp=spp, pps=1, depth=depth).render(rng=orc.RNG_COUNTER, seed=1, threads=4)
    d = np.sqrt(((film[...,:3].astype(np.float64)-ref[...,:3])**2).sum(-1))
    print("spp",spp,"depth",depth,"maxL2",d.max(),"ndiff",(d>0).sum(), "gpu",st['rays_closest'],st['rays_shadow'],st['rays_masked'],"cpu",ost['rays_closest'],ost['rays_shadow'],ost['rays_masked'])
    if d.max()>0:
        ys,xs = np.nonzero(d>0)
        for y,x in list(zip(ys,xs))[:5]: print("   ",y,x,film[y,x,:3],ref[y,x,:3])
