#!/bin/bash
# Run ON the GPU box: parity + work counts + A/B of the PHX_RUNAHEAD variant
R=${GRAFT_REPO_ROOT:-$(pwd)}
make -s -C $R/phosphorus_mk2_amd/csrc variant NAME=ra EXTRA="-DPHX_RUNAHEAD=1" || exit 1
make -s -C $R/phosphorus_mk2_amd/csrc variant NAME=racount EXTRA="-DPHX_RUNAHEAD=1 -DPHX_COUNT=1" || exit 1
PHX_LIB=$R/phosphorus_mk2_amd/libphx_hip_ra.so timeout -k 10 300 python3 -m pytest $R/tests/test_gpu_parity.py -x -q -k "render or random or stress or showroom or closures or glass" 2>&1 | tail -3 || exit 1
cp $R/phosphorus_mk2_amd/libphx_hip_count.so /tmp/keep_count.so
cp $R/phosphorus_mk2_amd/libphx_hip_racount.so $R/phosphorus_mk2_amd/libphx_hip_count.so
python3 $R/scripts/count_work.py > $R/gpurun_out/count_ra_100k.json; cat $R/gpurun_out/count_ra_100k.json
cp /tmp/keep_count.so $R/phosphorus_mk2_amd/libphx_hip_count.so
bash $R/scripts/ab.sh ra "base:" "ra:-DPHX_RUNAHEAD=1"
