#!/bin/bash
# same-box A/B of stage-1 LDS image layouts (pair offset / odd-row offset in pixels; 56/28 = linear 28-pixel stride)
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
PKG=transformer_cnn_hybrid_network_for_video_processing_amd
cp $PKG/build/conv_first_wave.o /tmp/orig.o
run() { python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline --no-pipeline --no-fwd-bwd-only 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   clips/s', round(d['value']), 'ms', round(d['ms_per_step'],4))"; }
for V in "56 28" "88 42" "104 50" "72 24" "56 28"; do
  set -- $V
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -I$PKG/csrc -Wno-unused-result -mllvm -amdgpu-mfma-vgpr-form -DS1W_PAIR_PX=$1 -DS1W_ODD_PX=$2 -c $PKG/csrc/conv_first_wave.hip -o $PKG/build/conv_first_wave.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $PKG/libhybrid_hip.so $PKG/build/*.o
  echo "== pair $1 odd $2"; run; run
done
cp /tmp/orig.o $PKG/build/conv_first_wave.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $PKG/libhybrid_hip.so $PKG/build/*.o
