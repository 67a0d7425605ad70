#!/usr/bin/env python3
"""Instruction budget of k_trace's streaming loop, read off the compiler's assembly.

    make -C phosphorus_mk2_amd/csrc asm && python scripts/ktrace_budget.py [--kernel k_traceILi1024ELb0] > profiles/r04_ktrace_budget.md

The loop of trace_stream (kernels.hip) compiles to a handful of basic-block regions that are recognised here by their
instructions, not by label numbers (those change from build to build):
  head      loop counter, ballot of idle lanes, refill decision                 (every iteration)
  refill    chunk bookkeeping, ranks of the idle lanes, ray fetch, reciprocals  (when >= refill_min lanes are idle)
  node      one node visit: stack push, child select / rank / address, fetch, origin decode, slab arithmetic, masks
  tri       one triangle test: select / rank / address, fetch, Moeller-Trumbore with the IEEE division, accept
  pop       pop the next group or finish the ray (hit record / radiance add)
  tail      exec-mask merges of the structured control flow back to the loop head
Every VALU instruction is priced with the issue costs MEASURED on gfx950 (profiles/r02_valu_ops.json, scripts/micro/valu_ops.hip):
VOP2 integer / fp32 add, sub, mul, fmac, and, or, shifts: 2 clocks per wave-instruction and SIMD; v_fma_f32 2.6; v_rcp_f32 8;
everything else (VOP3 encodings, conversions, min/max, compares, selects, bit-field ops) 4.  The table is an issue-time
estimate for ONE wave with all its lanes in the block — it says where the instructions are, not how long a launch takes.
"""
import argparse
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

TWO = {"v_mul_f32_e32", "v_add_f32_e32", "v_sub_f32_e32", "v_subrev_f32_e32", "v_fmac_f32_e32", "v_add_u32_e32", "v_sub_u32_e32", "v_subrev_u32_e32",
       "v_and_b32_e32", "v_or_b32_e32", "v_xor_b32_e32", "v_lshlrev_b32_e32", "v_lshrrev_b32_e32", "v_mov_b32_e32", "v_not_b32_e32", "v_min_u32_e32", "v_max_u32_e32"}


def clocks(op):
    if op in TWO:
        return 2.07
    if op in ("v_fma_f32",):
        return 2.61
    if op.startswith("v_rcp") or op.startswith("v_div_scale") or op.startswith("v_div_fmas") or op.startswith("v_div_fixup"):
        return 8.05 if op.startswith("v_rcp") else 4.1
    return 4.1


def kind(op):
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"):
        return "wait"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("global_") or op.startswith("flat_") or op.startswith("buffer_") or op.startswith("scratch_"):
        return "vmem"
    return "other"


def kernel_text(path, name):
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if l.startswith("_ZN3phx7" + name) and l.rstrip().endswith(":") or (l.startswith("_ZN3phx7" + name) and ": " in l))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    return lines[start:end + 1]


def instructions(lines):
    out = []
    for i, l in enumerate(lines):
        s = l.strip()
        if not s or s.startswith(";") or s.startswith(".") and not s.startswith(".LBB") or s.endswith(":") and not s.startswith(".LBB"):
            continue
        if s.startswith(".LBB"):
            out.append((i, "label", s.split(":")[0]))
            continue
        if s.startswith(";") or s.startswith("%"):
            continue
        op = s.split()[0]
        if re.match(r"^(v_|s_|ds_|global_|flat_|buffer_|scratch_)", op):
            out.append((i, op, s))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--asm", default=os.path.join(ROOT, "phosphorus_mk2_amd", "csrc", "kernels.s"))
    ap.add_argument("--kernel", default="k_traceILi1024ELb0")
    a = ap.parse_args()
    ins = instructions(kernel_text(a.asm, a.kernel))
    ops = [x for x in ins if x[1] != "label"]
    text = [x[2] for x in ops]

    def find(pred, start=0):
        for k in range(start, len(ops)):
            if pred(ops[k][1], ops[k][2]):
                return k
        raise SystemExit("marker not found")

    # the loop head: the watchdog counter (s_add_i32 sN, sN, -1 followed by s_cmp_lg_u32 sN, 0)
    head = find(lambda o, s: o == "s_add_i32" and s.endswith(", -1"))
    while not ops[head + 1][1].startswith("s_cmp_lg"):
        head = find(lambda o, s: o == "s_add_i32" and s.endswith(", -1"), head + 1)
    refill = find(lambda o, s: o == "s_bcnt1_i32_b64", head)           # popcount of the idle ballot
    ray_fetch = find(lambda o, s: o.startswith("v_rcp_f32"), refill)     # make_ray_ctx
    node = find(lambda o, s: o == "v_ffbh_u32_e32", ray_fetch)           # highest pending inner child
    node_start = node - 6                                                # the th == 0 && ng_hits > 0xffffff test in front of it
    first_cvt = find(lambda o, s: o.startswith("v_cvt_f32_ubyte"), node)
    fetch0 = find(lambda o, s: o == "ds_read_b128", node)
    fetch1 = max(k for k in range(fetch0, first_cvt) if ops[k][1] == "global_load_dwordx4")
    perm = find(lambda o, s: o == "ds_read_u8", first_cvt)
    last_slab = max(k for k in range(first_cvt, perm) if ops[k][1] == "v_alignbit_b32")
    node_end = perm + 4                                                  # ds_read_u8, bitop3, waitcnt, lshl, bitop3
    tri = find(lambda o, s: o == "v_ffbh_u32_e32", node_end)
    tri_fetch_end = find(lambda o, s: o == "global_load_dwordx2", tri)
    tri_end = find(lambda o, s: o == "ds_read_b64", tri)                 # the pop
    pop_end = find(lambda o, s: o == "global_store_dwordx3", tri_end)
    regions = [
        ("head: watchdog, idle ballot, refill decision", head, refill, "every iteration"),
        ("refill: chunk bookkeeping (LDS cursor, leader broadcast)", refill, ray_fetch - 30, "per refill"),
        ("refill: ranks of the idle lanes, ray fetch, reciprocals, octant", ray_fetch - 30, node_start, "per refill"),
        ("node: is-there-a-node test, highest pending child, stack push", node_start, fetch0 - 9, "per node-block execution"),
        ("node: child slot -> rank -> pool index -> LDS / global address", fetch0 - 9, fetch0, "per node-block execution"),
        ("node: fetch (4 x ds_read_b128, 4 x global_load_dwordx4 + address)", fetch0, fetch1 + 1, "per node-block execution"),
        ("node: decode (grid origin -> float, scale exponents, origin - o, x 1/d, near / far select)", fetch1 + 1, first_cvt, "per node-block execution"),
        ("node: slab arithmetic of the 8 children (cvt, fma, max3 / min3, pad fma, sub, or3, alignbit)", first_cvt, last_slab + 1, "per node-block execution"),
        ("node: valid / inner masks, octant permutation (LDS table), new group state", last_slab + 1, node_end + 1, "per node-block execution"),
        ("tri: pending test, triangle select -> rank -> address, fetch (3 loads)", node_end + 1, tri_fetch_end + 1, "per tri-block execution"),
        ("tri: Moeller-Trumbore incl. IEEE division, tie rule, accept", tri_fetch_end + 1, tri_end - 8, "per tri-block execution"),
        ("pop / finish: group done?, stack pop, hit record store, radiance read-modify-write", tri_end - 8, pop_end + 1, "per iteration (divergent)"),
    ]
    print("# k_trace instruction budget (`" + a.kernel + "`, from `kernels.s`)\n")
    print("Generated by `scripts/ktrace_budget.py` from the assembly of the committed sources (`make -C phosphorus_mk2_amd/csrc asm`).")
    print("Clocks = issue time of ONE wave on its SIMD with the measured per-instruction costs of `profiles/r02_valu_ops.json`.\n")
    print("| region | when | VALU | of which 4-clock | VALU clocks | SALU | branches | LDS | VMEM | waitcnt |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    tot = {}
    for name, lo, hi, when in regions:
        c = {"valu": 0, "salu": 0, "branch": 0, "lds": 0, "vmem": 0, "wait": 0, "other": 0}
        clk = 0.0; four = 0
        for k in range(lo, hi):
            kd = kind(ops[k][1]); c[kd] += 1
            if kd == "valu":
                cl = clocks(ops[k][1]); clk += cl; four += cl > 3.9
        print(f"| {name} | {when} | {c['valu']} | {four} | {clk:.0f} | {c['salu']} | {c['branch']} | {c['lds']} | {c['vmem']} | {c['wait']} |")
        grp = name.split(":")[0].split(" ")[0]
        t = tot.setdefault(grp, [0, 0.0])
        t[0] += c["valu"]; t[1] += clk
    print("\n| block | VALU instructions | VALU clocks |\n|---|---|---|")
    for g, (n, clk) in tot.items():
        print(f"| {g} | {n} | {clk:.0f} |")
    # the slab arithmetic by opcode
    print("\nSlab arithmetic by opcode (8 children):\n")
    hist = {}
    for k in range(first_cvt, last_slab + 1):
        if kind(ops[k][1]) == "valu":
            o = re.sub(r"_e(32|64)$", "", ops[k][1]); o = re.sub(r"ubyte[0-3]", "ubyteN", o)
            hist[o] = hist.get(o, 0) + 1
    print("| opcode | count | clocks each | clocks |\n|---|---|---|---|")
    for o, n in sorted(hist.items(), key=lambda x: -x[1]):
        cl = clocks(o if o in ("v_fma_f32",) else o + "_e32")
        print(f"| {o} | {n} | {cl:.2f} | {n * cl:.0f} |")


if __name__ == "__main__":
    main()
