cd /root/repo
mkdir -p gpurun_out/profiles_r05
python bench.py > gpurun_out/profiles_r05/r05_bench_c3.json 2> gpurun_out/profiles_r05/bench.err
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/profiles_r05/r05_bench_c3_driver_cmd.json 2>> gpurun_out/profiles_r05/bench.err
bash tools/profile_round.sh r05 > gpurun_out/profiles_r05/profile.log 2>&1
bash tools/timeline.sh r05 > gpurun_out/profiles_r05/timeline.log 2>&1
ls gpurun_out/profiles_r05 gpurun_out/timeline_r05 2>/dev/null
python - <<'PY'
import json
for f in ('r05_bench_c3.json','r05_bench_c3_driver_cmd.json'):
    d=json.loads(open('gpurun_out/profiles_r05/'+f).read().strip().splitlines()[-1])
    print(f, d['value'], d['ms_per_step'], d.get('fit_250_ms_per_cycle'), d.get('early_fit_ms_per_cycle'), d['organic_scene']['ms_per_step'], d['roofline']['frac'], d['roofline_lbs_projection']['frac'], d['roofline_lbs_projection']['full_unit'].get('frac_standalone'), d.get('speedup_vs_cpu_port'), d['kernel_us'])
PY
