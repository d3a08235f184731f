#!/bin/bash
# build scripts/micro/wgrad_bench and one library per variant:  wgrad_build.sh name:"flags" ...   (e.g. abl1:"-DHYB_ABL=1")
# (the sources with the timing-only ablation bits HYB_ABL live here, in scripts/micro/wgrad_variants/conv_wgrad_abl.hip + conv_wgrad_v3_abl.h:
#  copies of the round-3 product kernels; the product sources carry no ablation branches)
set -e
cd "$(dirname "$0")/../.."
PKG=transformer_cnn_hybrid_network_for_video_processing_amd
OUT=scripts/micro/wg
mkdir -p $OUT
HF="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -I$PKG/csrc -Iscripts/micro/wgrad_variants -DHYB_WGRAD_EXPERIMENTS -fno-slp-vectorize -Wno-unused-result"
hipcc $HF -c scripts/micro/wgrad_stub.hip -o $OUT/stub.o &
hipcc -O2 -std=c++17 scripts/micro/wgrad_bench.cpp -o $OUT/wgrad_bench -ldl &
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  ( hipcc $HF $flags -c scripts/micro/wgrad_variants/conv_wgrad_abl.hip -o $OUT/$name.o ) &
done
wait
for v in "$@"; do
  name=${v%%:*}
  hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libwg_$name.so $OUT/$name.o $OUT/stub.o
done
ls -la $OUT
