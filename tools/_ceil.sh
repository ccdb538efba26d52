cd /root/repo
export PYTHONPATH=/root/repo/scene-aware-3d-multi-human_amd
mkdir -p gpurun_out/ceil
for f in base ceil1 ceil4096 ceil32768; do
  echo "== $f"
  export MHHIP_LIB=/root/repo/variants/lib_$f.so
  python tools/raster_keys.py 2>&1 | grep "selection phase\|sha1" | cut -c1-120
  python tools/pair_stats.py 2>&1 | grep launches
  IN_CYCLE=1 python tools/pair_stats.py 2>&1 | grep launches
done > gpurun_out/ceil/stats.txt 2>&1
unset MHHIP_LIB
REPS=4 bash tools/ab_rotate.sh variants/lib_base.so variants/lib_ceil1.so variants/lib_ceil4096.so variants/lib_ceil32768.so > gpurun_out/ceil/ab.txt 2>&1
cat gpurun_out/ceil/stats.txt; tail -5 gpurun_out/ceil/ab.txt
