"""Closure-recipe baker: turns a material's shader graph — the BSDF node shaders of the reference with
constant inputs — into the flat closure list (`scenes.MaterialDesc` -> `phx_material`) the device consumes.

This is the offline stand-in for what happens per hit in the reference: OSL executes the material's shader
group and `material_t::details_t::eval_closure` (src/material.cpp:218-305) flattens the resulting closure
tree (MUL multiplies the colour weight down the tree, ADD visits A then B, `emission`/`background` assign
`result.e`, every other component becomes one `bsdf_t` lobe with weight = accumulated weight x component
weight).  With constant node inputs that tree does not depend on the hit, so it can be baked once.

Node semantics follow the shader sources (file:line = reference src/shaders/):
  diffuse_bsdf_node.osl:20-25       roughness == 0 -> Cs * diffuse(N) else Cs * oren_nayar(N, roughness)
  glossy_bsdf_node.osl:26-34        "sharp" or roughness == 0 -> Cs * reflection(N, 0)
                                    else Cs * microfacet(dist, N, 0, r*r, r*r, 0, 0)
  refraction_bsdf_node.osl:30-39    eta = IoR; sharp -> Cs * refraction(N, eta) else Cs * microfacet(dist, N, 0, r, r, eta, 1)
  sheen_bsdf_node.osl:20            Cs * sheen(N, roughness)
  transparent_bsdf.node.osl:14      Cs * transparent()
  diffuse_emitter_node.osl:18       (power / M_PI) * Cs * emission()
  background_node.osl:14            Cs * power * background()
  mix_closure_node.osl:20           A * (1 - fac) + B * fac
  add_node.osl:16                   A + B
The material description accepted by `bake_material` is the reference's YAML material schema
(src/codecs/scene/material.hpp:44-96): `shaders: [{name, layer, parameters: [{name, type, value}]}]`,
`connect: [{from: {slot, layer}, to: {slot, layer}}]`; the LAST layer is the group's root (OSL convention).
  fresnel_dielectric_node.osl:16-20 out = fresnel_dielectric(dot(I, N), backfacing ? 1/max(1e-5, IoR) : max(1e-5, IoR))
All arithmetic is fp32, as in OSL.  Closures multiplied by an all-zero weight are dropped (OSL returns a null
closure for `closure * 0`).  ONE hit-dependent input is supported, the one the reference's own Blender exporter produces:
`fresnel_dielectric_node.out` driving `mix_closure_node.fac` (Blender's glass node = mix(refraction, glossy, fresnel),
plugins/blender/blender/shader.hpp:306-335).  It cannot be baked into a number, so the recipe records it: the closures under
that mix carry `fac_mode` / `fac_ior` / `pre_weight` and the device (bsdf.h: material_at_hit) and the oracle evaluate the factor
at every hit.  Other hit-dependent inputs (textures, noise, normal maps) cannot be expressed and raise.
"""
import math

import numpy as np

from . import abi
from .scenes import LobeDesc, MaterialDesc

f32 = np.float32
M_PI = f32(math.pi)  # OSL's M_PI is a float


def _color(v):
    if isinstance(v, dict):  # YAML {type: rgb, value: [r,g,b]}
        v = v["value"]
    if np.isscalar(v):
        return np.array([v, v, v], f32)
    a = np.asarray(v, f32)
    assert a.shape == (3,)
    return a


class Comp:
    """a closure component: id = bsdf_t::type_t (src/bsdf.hpp:14-24) + its parameter struct (src/bsdf/params.hpp)"""
    def __init__(self, cid, **params):
        self.cid, self.params = cid, params


class Mul:
    def __init__(self, weight, closure):
        self.weight, self.closure = _color(weight), closure


class Fac:
    """the output of a fresnel_dielectric_node: a float known only at the hit"""
    def __init__(self, ior):
        self.ior = f32(ior)


class MulFac:
    """closure * fac (mode FAC_MIX_B) or closure * (1 - fac) (mode FAC_MIX_A) with fac a Fac"""
    def __init__(self, mode, fac, closure):
        self.mode, self.fac, self.closure = mode, fac, closure


class Add:
    def __init__(self, a, b):
        self.a, self.b = a, b


def mul(weight, closure):
    w = _color(weight)
    if closure is None or not w.any():
        return None  # OSL: closure * 0 is the null closure
    return Mul(w, closure)


def add(a, b):
    if a is None:
        return b
    if b is None:
        return a
    return Add(a, b)


# ---- the node shaders ----------------------------------------------------------------------------------
def diffuse_bsdf_node(Cs=1.0, roughness=0.0, **_):
    r = f32(roughness)
    return mul(Cs, Comp(abi.LOBE_DIFFUSE) if r == 0 else Comp(abi.LOBE_OREN_NAYAR, alpha=r))


def glossy_bsdf_node(distribution="ggx", Cs=1.0, roughness=0.0, **_):
    r = f32(roughness)
    r2 = f32(r * r)
    if distribution == "sharp" or r == 0:
        return mul(Cs, Comp(abi.LOBE_REFLECTION, eta=f32(0)))
    return mul(Cs, Comp(abi.LOBE_MICROFACET, distribution=distribution, xalpha=r2, yalpha=r2, eta=f32(0), refract=0))


def refraction_bsdf_node(distribution="ggx", Cs=1.0, IoR=0.5, roughness=0.0, **_):
    r = f32(roughness)
    eta = f32(IoR)  # "backfacing() ? 1/f : f" is commented out in the shader
    if distribution == "sharp" or r == 0:
        return mul(Cs, Comp(abi.LOBE_REFRACTION, eta=eta))
    return mul(Cs, Comp(abi.LOBE_MICROFACET, distribution=distribution, xalpha=r, yalpha=r, eta=eta, refract=1))


def sheen_bsdf_node(Cs=1.0, roughness=0.0, **_):
    return mul(Cs, Comp(abi.LOBE_SHEEN, r=f32(roughness)))


def transparent_bsdf_node(Cs=1.0, **_):
    return mul(Cs, Comp(abi.LOBE_TRANSPARENT))


def diffuse_emitter_node(power=1.0, Cs=1.0, **_):
    return mul(f32(f32(power) / M_PI) * _color(Cs), Comp(abi.LOBE_EMISSIVE))


def background_node(Cs=0.0, power=1.0, **_):
    return mul(_color(Cs) * f32(power), Comp(abi.LOBE_BACKGROUND))


def fresnel_dielectric_node(IoR=1.45, **_):
    return Fac(IoR)


def mix_closure_node(A=None, B=None, fac=0.5, **_):
    if isinstance(fac, Fac):  # Cout = A * (1 - fac) + B * fac with fac evaluated per hit
        return add(MulFac(abi.FAC_MIX_A, fac, A) if A is not None else None, MulFac(abi.FAC_MIX_B, fac, B) if B is not None else None)
    if not np.isscalar(fac):
        raise ValueError("mix_closure_node.fac is driven by a node this baker cannot express (hit-dependent)")
    fac = f32(fac)
    return add(mul(f32(f32(1) - fac), A), mul(fac, B))


def add_node(A=None, B=None, **_):
    return add(A, B)


NODES = {f.__name__: f for f in (diffuse_bsdf_node, glossy_bsdf_node, refraction_bsdf_node, sheen_bsdf_node, transparent_bsdf_node,
                                 diffuse_emitter_node, background_node, mix_closure_node, add_node, fresnel_dielectric_node)}
UNBAKEABLE = {"fresnel_node", "texture_node", "normal_map_node", "random_noise_2d_node", "random_noise_3d_node",
              "musgrave_noise_3d_node", "environment_node", "mix_color_node", "blackbody_node"}


# ---- material.cpp:218-305 -------------------------------------------------------------------------------
def flatten(tree):
    """eval_closure: closure tree -> MaterialDesc (lobes in visiting order, e = last emission/background weight).  The tree is
    walked as material.cpp:218-305 walks it: MUL multiplies the weight down, ADD visits A then B.  A Fresnel-driven factor
    splits a closure's weight into the constant part above it (pre_weight), the factor itself (fac_mode, fac_ior) and the
    constant part below it (weight): at a hit the weight is (pre_weight * term) * weight, the same order of multiplications."""
    lobes, state = [], {"e": (0.0, 0.0, 0.0), "emitter": False}

    def visit(c, w, fac=None):
        # w: the constant weight accumulated so far BELOW the hit-dependent factor (or all of it when there is none);
        # fac = (mode, ior, pre): the factor met on the way down and the constant weight accumulated ABOVE it
        if c is None:
            return
        if isinstance(c, Mul):
            visit(c.closure, (w * c.weight).astype(f32), fac)
        elif isinstance(c, MulFac):
            if fac is not None:
                raise ValueError("a Fresnel-driven mix below another one: two hit-dependent factors on one closure are not supported")
            visit(c.closure, np.ones(3, f32), (c.mode, c.fac.ior, w))
        elif isinstance(c, Add):
            visit(c.a, w, fac)
            visit(c.b, w, fac)
        else:
            if c.cid in (abi.LOBE_EMISSIVE, abi.LOBE_BACKGROUND):
                if fac is not None:
                    raise ValueError("emission under a Fresnel-driven mix is not supported")
                state["e"] = tuple(float(x) for x in w)  # assignment: a later emission overwrites an earlier one
                state["emitter"] = state["emitter"] or c.cid == abi.LOBE_EMISSIVE  # material.cpp:205-211
                return
            p = c.params
            if c.cid == abi.LOBE_MICROFACET and p.get("distribution", "ggx") not in ("ggx", "beckmann"):
                raise ValueError(f"unsupported distribution {p['distribution']!r} (src/bsdf.cpp:53-71)")
            extra = {}
            if fac is not None:
                extra = {"fac_mode": int(fac[0]), "fac_ior": float(fac[1]), "pre_weight": tuple(float(x) for x in fac[2])}
            lobes.append(LobeDesc(c.cid, tuple(float(x) for x in w), alpha=float(p.get("alpha", 0.0)), eta=float(p.get("eta", 0.0)),
                                  xalpha=float(p.get("xalpha", 0.0)), yalpha=float(p.get("yalpha", 0.0)), refract=int(p.get("refract", 0)),
                                  r=float(p.get("r", 0.0)), **extra))
    visit(tree, np.ones(3, f32))
    if len(lobes) > abi.MAX_LOBES:
        raise ValueError(f"{len(lobes)} lobes: bsdf_t holds at most {abi.MAX_LOBES} (src/bsdf.hpp:9)")
    return MaterialDesc(lobes=lobes, emission=state["e"], is_emitter=state["emitter"])


def bake_material(desc):
    """`desc`: one entry of the reference's YAML `materials:` map (already parsed, e.g. by yaml.safe_load)."""
    layers, order = {}, []
    for sh in desc["shaders"]:
        name, layer = sh["name"], sh["layer"]
        if name in UNBAKEABLE:
            raise ValueError(f"shader {name!r} depends on the hit (texture / noise / view direction): not a constant closure recipe")
        if name not in NODES:
            raise ValueError(f"unknown shader {name!r}")
        params = {}
        for p in sh.get("parameters", []) or []:
            t = p["type"]
            if t == "float":
                params[p["name"]] = float(p["value"])
            elif t == "rgb":
                params[p["name"]] = _color(p["value"])
            elif t == "string":
                params[p["name"]] = str(p["value"])
            else:
                raise ValueError("Unknown parameter type: " + t)  # material.hpp:79
        layers[layer] = (NODES[name], params)
        order.append(layer)
    edges = {}
    for e in desc.get("connect", []) or []:
        edges.setdefault(e["to"]["layer"], []).append((e["to"]["slot"], e["from"]["layer"], e["from"]["slot"]))
    cache = {}

    def evaluate(layer):
        if layer not in cache:
            fn, params = layers[layer]
            args = dict(params)
            for slot, src_layer, src_slot in edges.get(layer, []):
                if src_slot not in ("Cout", "out") or (src_slot == "out" and layers[src_layer][0] is not fresnel_dielectric_node):
                    raise ValueError(f"connection from {src_layer}.{src_slot}: only closure outputs (Cout) and fresnel_dielectric_node.out can be expressed")
                args[slot] = evaluate(src_layer)
            cache[layer] = fn(**args)
        return cache[layer]
    return flatten(evaluate(order[-1]))


def bake_materials(yaml_materials):
    """name -> MaterialDesc for a whole `materials:` map; ids follow insertion order (scene_t::add, src/scene.cpp:84-90)."""
    return {name: bake_material(d) for name, d in yaml_materials.items()}
