#!/bin/bash
# developer tool: variants/lib_<name>.so = the whole library recompiled with extra flags (for macros that live in shared headers,
# e.g. -DRG_UNIT=2048)        usage: tools/mkvariant_all.sh <name> [flags...]
set -e
cd /root/repo
name=$1; shift
P=scene-aware-3d-multi-human_amd
mkdir -p variants /tmp/mkva_$name
objs=""
for f in $P/csrc/*.hip; do
  o=/tmp/mkva_$name/$(basename ${f%.hip}).o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -munsafe-fp-atomics -fno-slp-vectorize -fno-vectorize "$@" -c $f -o $o &
  objs="$objs $o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o variants/lib_$name.so
echo variants/lib_$name.so
