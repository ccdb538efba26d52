"""developer tool: one line of a bench.py JSON line read from stdin (label, headline ms, organic ms, fit ms per cycle): used by
one-line A/B loops through gpurun"""
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
ph = (d.get('fit_250') or {}).get('phases') or {}
print(sys.argv[1], d['ms_per_step'], (d.get('organic_scene') or {}).get('ms_per_step'), d.get('fit_250_ms_per_cycle'), ph.get('cycles_1_29_ms'), ph.get('cycles_30_59_ms'), ph.get('cycles_60_248_ms'))
