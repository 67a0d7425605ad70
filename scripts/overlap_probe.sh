#!/bin/bash
# Run ON the GPU box: do two device objects on ONE GPU (each with its own stream, half of the tiles each, one shared tile queue)
# overlap one's memory-bound shade launches with the other's ALU-bound k_trace?  PHX_TRACE_WG_CAP caps k_trace's resident workgroups
# per CU so that the other stream's kernels find wave slots.
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=${1:-100000}
echo "one device:"; PROBE_DEVICES=1 python3 $R/scripts/two_stream_probe.py $T
for cfg in "PHX_TRACE_WG_CAP=0" "PHX_TRACE_WG_CAP=1" "PHX_TRACE_BLOCK=512 PHX_TRACE_WG_CAP=3" "PHX_TRACE_BLOCK=512 PHX_TRACE_WG_CAP=2" "PHX_TRACE_BLOCK=256 PHX_TRACE_WG_CAP=7" "PHX_TRACE_BLOCK=256 PHX_TRACE_WG_CAP=6"; do
  echo "two devices, $cfg:"; env $cfg PROBE_DEVICES=2 python3 $R/scripts/two_stream_probe.py $T
done
echo "three devices, no cap:"; PROBE_DEVICES=3 python3 $R/scripts/two_stream_probe.py $T
