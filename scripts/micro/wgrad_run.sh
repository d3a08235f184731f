#!/bin/bash
# run every built variant (scripts/micro/wg/libwg_*.so) through wgrad_bench; first argument = reference variant for the result compare
cd "$(dirname "$0")/wg"
REF=${1:-}
for f in libwg_*.so; do
  echo "== $f"
  if [ -n "$REF" ] && [ "$f" != "libwg_$REF.so" ]; then timeout -k 10 120 ./wgrad_bench ./$f ./libwg_$REF.so || exit 1
  else timeout -k 10 120 ./wgrad_bench ./$f || exit 1; fi
done
