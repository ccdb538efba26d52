#!/bin/bash
# developer tool: variants/lib_<name>.so = the tree's library with ONE source recompiled with extra flags
# usage: tools/mkvariant.sh <name> <source.hip> [flags...]   (the other objects come from scene-aware-3d-multi-human_amd/build)
set -e
cd /root/repo
name=$1; src=$2; shift 2
P=scene-aware-3d-multi-human_amd
mkdir -p variants /tmp/mkv_$name
python -c "import sys; sys.path.insert(0,'$P'); from mhhip import build; build.build()" >/dev/null
extra="-fno-slp-vectorize -fno-vectorize"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -munsafe-fp-atomics $extra "$@" -c ${SRCFILE:-$P/csrc/$src} -o /tmp/mkv_$name/${src%.hip}.o
objs=""
for f in $P/build/*.o; do
  b=$(basename $f)
  if [ "$b" = "${src%.hip}.o" ]; then objs="$objs /tmp/mkv_$name/$b"; else objs="$objs $f"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o variants/lib_$name.so
echo variants/lib_$name.so
