#!/bin/bash
# A/B of stage-1 build variants on the GPU box: recompiles conv_first_wave.o with extra flags, relinks, and times the stage alone
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
PKG=transformer_cnn_hybrid_network_for_video_processing_amd
for V in "-DS1W_SB_UNROLL=2 -DS1W_MIN_WAVES=4" "-DS1W_SB_UNROLL=1 -DS1W_MIN_WAVES=4" "-DS1W_SB_UNROLL=1 -DS1W_MIN_WAVES=5" "-DS1W_SB_UNROLL=2 -DS1W_MIN_WAVES=5" "-DS1W_SB_UNROLL=1 -DS1W_MIN_WAVES=6"; do
  echo "== $V"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -I$PKG/csrc -Wno-unused-result -mllvm -amdgpu-mfma-vgpr-form $V -c $PKG/csrc/conv_first_wave.hip -o $PKG/build/conv_first_wave.o 2>&1 | grep -E "error|spill" || true
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $PKG/libhybrid_hip.so $PKG/build/*.o
  for W in 1024 2048; do HYB_S1_FWD_WGS=$W timeout -k 10 100 python scripts/s1_bench.py 2>/dev/null | cut -c1-90; done
done
