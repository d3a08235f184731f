#!/bin/bash
P=transformer_cnn_hybrid_network_for_video_processing_amd/libhybrid_hip.so
cp $P /tmp/orig.so
echo "== baseline"; python scripts/bench_kernels.py | grep -E "conv[234]_(fwd|dgrad)"
for k in 4 5; do cp gpurun_dbg_$k.so $P; echo "== variant $k"; python scripts/bench_kernels.py | grep -E "conv[234]_(fwd|dgrad)"; done
cp /tmp/orig.so $P
