cd /root/repo
for w in 1 0 1 0; do echo "winners=$w"; MHHIP_RASTER_WINNERS=$w python tools/fit_cycles.py 2>&1 | grep -A1 "^fit(250)" | tail -2; done
