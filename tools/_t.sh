cd /root/repo
export PYTHONPATH=/root/repo/scene-aware-3d-multi-human_amd
python tools/raster_keys.py 2>&1 | grep "sha1" | cut -c1-80
REPS=4 bash tools/ab_rotate.sh default variants/lib_e0q0.so variants/lib_e1q0.so 2>&1 | tail -3
