#!/bin/bash
# Run ON the GPU box: the parity gate at full size on every workload class (device film vs CPU oracle, pixel by pixel).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
python3 $R/scripts/full_parity.py 256 100000 auto   | tail -1 > $O/full_parity_100k_256spp.json   && echo "100k done" &&
python3 $R/scripts/full_parity.py 64 1000000 auto   | tail -1 > $O/full_parity_1M_64spp.json      && echo "1M host done" &&
python3 $R/scripts/full_parity.py 64 1000000 device | tail -1 > $O/full_parity_1M_64spp_device.json && echo "1M device done" &&
python3 $R/scripts/full_parity.py 64 cornell auto   | tail -1 > $O/full_parity_cornell_64spp.json && echo "cornell done" &&
python3 $R/scripts/full_parity.py 32 zoo:500000 auto | tail -1 > $O/full_parity_zoo_32spp.json    && echo "zoo done" &&
python3 $R/scripts/full_parity.py 64 showroom:200000 auto | tail -1 > $O/full_parity_showroom_64spp.json && echo "showroom done" &&
python3 $R/scripts/full_parity.py 32 bmwroom:500000 auto | tail -1 > $O/full_parity_bmwroom_32spp.json && echo "bmwroom done"
