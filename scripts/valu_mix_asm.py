#!/usr/bin/env python3
"""Does scripts/micro/valu_mix.hip time the node test k_trace runs?  

Compiles the micro-benchmark to gfx950 assembly, reads the inner loops of its node-test kernel (MODE 0) and of its skeleton
(MODE 2: the same operand perturbation, no test) and compares

    VALU instructions of the priced test = loop(MODE 0) - loop(MODE 2)
    v_cvt_f32_ubyteN in loop(MODE 0)

with the node test inside k_trace<1024, false> in csrc/kernels.s (`make -C phosphorus_mk2_amd/csrc asm`): the instructions from
the last of the node's four global loads to the octant-table lookup that ends node_hitmask — origin decode, scale exponents,
near / far selects, the slab arithmetic of the eight children, the valid / inner masks — and likewise for the triangle test (MODE 1
minus MODE 3 against k_trace's Moeller-Trumbore, loads to the |det| compare).  Asserts 48 conversions in both and a VALU count
within 5 % (`--tolerance`; triangle test: twice that), and prints one JSON object (merged into profiles/rNN_valu_mix.json by valu_mix_json.py).

    python scripts/valu_mix_asm.py [--no-assert]
"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "phosphorus_mk2_amd", "csrc")
INSTR = re.compile(r"^\s+((?:v_|s_|ds_|global_|flat_|buffer_|scratch_)\S+)")


def kernel_lines(path, mangled_prefix):
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if l.startswith(mangled_prefix) and ":" in l)
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    return lines[start:end + 1]


def ops_of(lines):
    out = []
    for l in lines:
        m = INSTR.match(l)
        if m:
            out.append(m.group(1))
        elif l.startswith(".LBB"):
            out.append("label:" + l.split(":")[0])
    return out


def loops(lines):
    """the instruction lines of every loop of a kernel, keyed by its header: the compiler annotates the header block ("Loop Header") and
    every other block of the loop ("in Loop: Header=BBn_m"); a loop need not end in a branch back to its header (rotated loops fall into it)"""
    out = {}
    cur = None
    for l in lines:
        if l.startswith(".LBB"):
            name = l.split(":")[0].lstrip(".L")
            m = re.search(r"in Loop: Header=(BB\w+)", l)
            cur = name if "Loop Header" in l else (m.group(1) if m else None)
            if cur is not None:
                out.setdefault(cur, [])
            continue
        if cur is not None:
            out[cur].append(l)
    return out


def count(lines):
    ops = [o for o in ops_of(lines) if not o.startswith("label:")]
    valu = [o for o in ops if o.startswith("v_")]
    return {"valu": len(valu), "cvt_ubyte": sum(o.startswith("v_cvt_f32_ubyte") for o in valu), "fma": sum(o.startswith("v_fma_f32") or o.startswith("v_fmac_f32") for o in valu),
            "salu": sum(o.startswith("s_") and not o.startswith("s_waitcnt") and not o.startswith("s_nop") and not o.startswith("s_cbranch") for o in ops),
            "lds": sum(o.startswith("ds_") for o in ops), "vmem": sum(o.startswith("global_") for o in ops)}


def micro_loops(asm):
    res = {}
    for mode in (0, 1, 2, 3):
        kl = kernel_lines(asm, "_Z1kILi%dEEv" % mode)
        res[mode] = count(max(loops(kl).values(), key=len))  # the timed loop is the kernel's longest
    return res


def ktrace_node_test(asm, kernel="_ZN3phx7k_traceILi1024ELb0E"):
    kl = kernel_lines(asm, kernel)
    first_cvt = next(i for i, l in enumerate(kl) if "v_cvt_f32_ubyte" in l)
    start = max(i for i in range(first_cvt) if "global_load_dwordx4" in kl[i]) + 1
    lut = next(i for i in range(first_cvt, len(kl)) if re.match(r"^\s+ds_read_u8", kl[i]))
    # the permuted inner mask is merged into the hit mask by the next two VALU instructions after the table read
    end = lut
    seen = 0
    while seen < 2:
        end += 1
        if INSTR.match(kl[end]) and INSTR.match(kl[end]).group(1).startswith("v_"):
            seen += 1
    node = count(kl[start:end + 1])
    # the triangle test: from the last of the record's three loads to the |det| > 1e-8 compare that closes mt_intersect (the accept's
    # five register moves and the lane's group bookkeeping behind it are the loop's, not the test's)
    tload = next(i for i in range(end, len(kl)) if "global_load_dwordx2" in kl[i])
    tend = next(i for i in range(tload, len(kl)) if re.match(r"^\s+v_cmp_gt_f32\S*\s+.*\|v\d+\|", kl[i]))
    return node, count(kl[tload + 1:tend + 1])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--no-assert", action="store_true")
    ap.add_argument("--tolerance", type=float, default=0.05)
    ap.add_argument("--kernels-asm", default=os.path.join(CSRC, "kernels.s"))
    a = ap.parse_args()
    if not os.path.exists(a.kernels_asm):
        subprocess.run(["make", "-C", CSRC, "asm"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "valu_mix.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "--offload-arch=gfx950", "-I", CSRC, "-w",
                        "-S", "--cuda-device-only", "-o", out, os.path.join(ROOT, "scripts", "micro", "valu_mix.hip")], check=True, stderr=subprocess.DEVNULL)
        micro = micro_loops(out)
    kt, kt_tri = ktrace_node_test(a.kernels_asm)
    priced = micro[0]["valu"] - micro[2]["valu"]
    rec = {"micro_loop_node": micro[0], "micro_loop_skeleton": micro[2], "micro_loop_tri": micro[1], "micro_loop_skeleton_tri": micro[3],
           "priced_node_test_valu": priced, "priced_node_test_cvt_ubyte": micro[0]["cvt_ubyte"] - micro[2]["cvt_ubyte"],
           "priced_tri_test_valu": micro[1]["valu"] - micro[3]["valu"],
           "ktrace_node_test": kt, "valu_ratio_micro_over_ktrace": priced / kt["valu"],
           "ktrace_tri_test": kt_tri, "tri_valu_ratio_micro_over_ktrace": (micro[1]["valu"] - micro[3]["valu"]) / kt_tri["valu"]}
    print(json.dumps(rec))
    if not a.no_assert:
        assert rec["priced_node_test_cvt_ubyte"] == 48 and kt["cvt_ubyte"] == 48, rec
        assert abs(rec["valu_ratio_micro_over_ktrace"] - 1.0) <= a.tolerance, rec
        assert abs(rec["tri_valu_ratio_micro_over_ktrace"] - 1.0) <= 2 * a.tolerance, rec


if __name__ == "__main__":
    main()
