cd /root/repo
export PYTHONPATH=/root/repo/scene-aware-3d-multi-human_amd
MHHIP_SIDE_SPLIT=1 python -m pytest tests/test_fit_full_gpu.py tests/test_optimizer_raster_gpu.py -q -x 2>&1 | tail -3
REPS=5 bash tools/ab_rotate.sh default "MHHIP_SIDE_SPLIT=1" 2>&1 | tail -2
