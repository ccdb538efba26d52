mkdir -p gpurun_out/fuzz5
run() { name=$1; shift; ( timeout 600 env "$@" python tools/$name.py > gpurun_out/fuzz5/$name.$RANDOM.log 2>&1; echo "$name $* rc=$?" >> gpurun_out/fuzz5/summary.txt ); }
run fuzz_raster SEED=9101 CASES=120
run fuzz_raster_grads SEED=9102 CASES=100
run fuzz_cycle SEED=9103 CASES=40 F64=both
run fuzz_cycle SEED=9104 CASES=30 EDGE=1 F64=both
run fuzz_kept SEED=9105 CASES=60
run fuzz_lbs SEED=9106 CASES=40
run fuzz_determinism SEED=9107 CASES=12
run fuzz_graphs SEED=9108 CASES=6
run fuzz_raster_closeup SEED=9109 CASES=12
cat gpurun_out/fuzz5/summary.txt
