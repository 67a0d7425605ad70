import sys, numpy as np
sys.path.insert(0, '.')
from phosphorus_mk2_amd import scenes, xpu
from oracle import oracle as orc
sc = scenes.cornell(64, 64)
sc.materials.append(scenes.MaterialDesc(lobes=[], emission=(0.3, 0.4, 0.5)))
sc.environment_material = len(sc.materials) - 1
sc.meshes = sc.meshes[:1] + sc.meshes[5:]
for normals in (False, True):
  for spp, depth in [(1,1),(1,2),(4,9)]:
    film, st = xpu.render(sc, spp=spp, pps=1, depth=depth, seed=2, normals=normals)
    res = orc.Oracle(sc, spp=spp, pps=1, depth=depth).render(rng=orc.RNG_COUNTER, seed=2, threads=4, normals=normals)
    ref, ost = res[0], res[1]
    d = np.sqrt(((film[...,:3].astype(np.float64)-ref[...,:3])**2).sum(-1))
    print("normals",normals,"spp",spp,"depth",depth,"maxL2",d.max(),"ndiff",(d>0).sum(), "gpu",st['rays_closest'],st['rays_shadow'],st['rays_masked'],"cpu",ost['rays_closest'],ost['rays_shadow'],ost['rays_masked'])
    if d.max()>0:
        ys,xs = np.nonzero(d>0)
        for y,x in list(zip(ys,xs))[:5]: print("   ",y,x,film[y,x,:3],ref[y,x,:3])
    if normals:
        dn = np.abs(film[...,4:7]-res[2]).max(); print("   normals diff", dn)
