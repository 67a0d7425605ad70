#!/usr/bin/env python3
"""Hash of the source text of k_trace's priced arithmetic: the functions of csrc/bvh8.h that scripts/micro/valu_mix.hip times and
bench.py's roofline prices a frame's node visits and triangle tests with.  The micro-benchmark embeds the hash of the sources it
was BUILT from in its JSON line (profiles/r*_valu_mix.json); bench.py recomputes it from the tree and refuses a peak whose hash
differs (the round-4 verdict: the r02 peak had outlived three edits of the node test).

    python scripts/src_hash.py            -> the hash of the tree's bvh8.h
"""
import hashlib
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BVH8 = os.path.join(ROOT, "phosphorus_mk2_amd", "csrc", "bvh8.h")
# what a node test and a triangle test are made of (valu_mix.hip calls node_hitmask and mt_intersect with a ray from make_ray_ctx)
PRICED_FUNCTIONS = ("node_origin_decode", "mt_intersect", "ray_rcp", "make_ray_ctx", "perm_xor8", "node_hit8", "node_hitmask")


def function_text(src, name):
    """every definition of `name` in src: from the line that declares it to its closing brace (brace matching on code with no
    braces inside strings or comments, which holds for bvh8.h)"""
    out = []
    for m in re.finditer(r"^[^\n/]*\b%s\s*\([^;{]*\)\s*\{" % re.escape(name), src, re.M):
        i = m.end() - 1
        depth = 0
        while True:
            c = src[i]
            depth += c == "{"
            depth -= c == "}"
            i += 1
            if depth == 0:
                break
        out.append(src[m.start():i])
    return out


def priced_source_hash(path=BVH8):
    src = open(path).read()
    h = hashlib.sha256()
    for name in PRICED_FUNCTIONS:
        bodies = function_text(src, name)
        if not bodies:
            raise RuntimeError(f"{name} not found in {path}")
        for b in bodies:
            h.update(name.encode() + b"\0" + re.sub(r"[ \t]+", " ", b).encode() + b"\0")
    # the switches that select the code path inside those functions
    for m in re.finditer(r"^#define (PHX_(?:FAST_RCP|PACKED_FMA|SIGN_ACCUM|NO_NEG_ZERO|PAD_FMA|BITOP3|F16_PLANES|GRID_BITS))\b[ \t]*([^\n/]*)", src, re.M):
        h.update((m.group(1) + "=" + m.group(2).strip()).encode() + b"\0")
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(priced_source_hash(sys.argv[1] if len(sys.argv) > 1 else BVH8))
