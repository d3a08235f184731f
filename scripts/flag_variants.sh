#!/bin/bash
# same-box A/B of per-file compiler flags: recompile one object with extra flags, relink, run the bench; restore at the end
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
PKG=transformer_cnn_hybrid_network_for_video_processing_amd
FILE=$1; shift
cp $PKG/build/${FILE%.hip}.o /tmp/orig.o
run() { python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline --no-pipeline --no-fwd-bwd-only 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   clips/s', round(d['value']), 'ms', round(d['ms_per_step'],4))"; }
echo "== baseline"; run; run
for V in "$@"; do
  echo "== $FILE $V"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -I$PKG/csrc -Wno-unused-result $V -c $PKG/csrc/$FILE -o $PKG/build/${FILE%.hip}.o 2>&1 | grep -E "error|spill" || true
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $PKG/libhybrid_hip.so $PKG/build/*.o
  run; run
done
cp /tmp/orig.o $PKG/build/${FILE%.hip}.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $PKG/libhybrid_hip.so $PKG/build/*.o
echo "== baseline again"; run
