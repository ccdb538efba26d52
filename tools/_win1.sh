cd /root/repo
export PYTHONPATH=/root/repo/scene-aware-3d-multi-human_amd
for w in 1 0; do for t in 1 2 3; do echo "winners=$w timing=$t"; MHHIP_RASTER_WINNERS=$w R_TIMING=$t MHHIP_LIB=/root/repo/variants/lib_t$t.so IN_CYCLE=1 python tools/pair_stats.py 2>&1 | grep "timing build"; done; done
