#!/bin/bash
# Run ON the GPU box: the VALU peak behind bench.py's roofline, from TODAY's bvh8.h.
#   1. scripts/micro/valu_mix            -> node tests / triangle tests per second (+ the hash of the sources it was built from)
#   2. rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE on the same binary -> the shader clock during the node-test kernel
#      (GRBM_GUI_ACTIVE is summed over the 8 XCDs: cycles / 8 / kernel time)
#   3. scripts/micro/valu_ops            -> issue cost per instruction
#   0. scripts/valu_mix_asm.py           -> the instruction counts of the micro-benchmark's loops against k_trace's own node / triangle test
# scripts/valu_mix_json.py merges 1 + 2 into gpurun_out/valu_mix.json (committed as profiles/rNN_valu_mix.json).
R=${GRAFT_REPO_ROOT:-$(pwd)}
make -C $R/scripts/micro valu_mix valu_ops valu_ops2 > /dev/null 2>&1 || { echo "micro build failed"; exit 1; }
OUT=$R/gpurun_out/valu_mix
rm -rf $OUT; mkdir -p $OUT
# 0. is the micro-benchmark's loop the node test k_trace runs?  (48 v_cvt_f32_ubyteN, VALU count within 5 % of the kernel's: asserted)
make -C $R/phosphorus_mk2_amd/csrc asm > /dev/null 2>&1 || { echo "make asm failed"; exit 1; }
python3 $R/scripts/valu_mix_asm.py > $OUT/asm_check.json || { echo "valu_mix_asm.py: the micro-benchmark does not time k_trace's node test"; cat $OUT/asm_check.json; exit 1; }
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 $R/scripts/micro/valu_mix > $OUT/valu_mix.out 2> $OUT/valu_mix.err || { echo "valu_mix failed"; cat $OUT/valu_mix.err; exit 1; }
timeout -k 10 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/grbm -- $R/scripts/micro/valu_mix > $OUT/grbm.log 2>&1; echo "grbm rc=$?"
timeout -k 10 200 $R/scripts/micro/valu_ops > $OUT/valu_ops.json 2> $OUT/valu_ops.err; echo "valu_ops rc=$?"
timeout -k 10 200 $R/scripts/micro/valu_ops2 > $OUT/valu_ops2.json 2> $OUT/valu_ops2.err; echo "valu_ops2 rc=$?"
python3 $R/scripts/valu_mix_json.py $OUT > $R/gpurun_out/valu_mix.json && cat $R/gpurun_out/valu_mix.json
