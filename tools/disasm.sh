#!/bin/bash
# developer tool: disassemble one kernel of the built library into /tmp/mhdis/<kernel>.s   (usage: tools/disasm.sh k_raster_strip)
set -e
K=${1:-k_raster_strip}
D=/tmp/mhdis
mkdir -p $D
cp "$(dirname "$0")/../scene-aware-3d-multi-human_amd/mhhip/libmhmocap_hip.so" $D/lib.so
(cd $D && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading lib.so > /dev/null 2>&1)
for f in $D/lib.so.*gfx950; do
  if /opt/rocm/lib/llvm/bin/llvm-objdump -t $f | grep -q "$K"; then
    /opt/rocm/lib/llvm/bin/llvm-objdump -d $f | awk -v k="$K" '/^[0-9a-f]+ <.*>:/{f = index($0, k) > 0} f' > $D/$K.s
  fi
done
wc -l $D/$K.s
