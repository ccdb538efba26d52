mkdir -p gpurun_out/fuzz9
run() { name=$1; shift; ( timeout 600 env "$@" python tools/$name.py > gpurun_out/fuzz9/$name.$RANDOM.log 2>&1; echo "$name $* rc=$?" >> gpurun_out/fuzz9/summary.txt ); }
run fuzz_raster SEED=9901 CASES=120
run fuzz_raster_grads SEED=9902 CASES=100
run fuzz_cycle SEED=9903 CASES=40 F64=both
run fuzz_cycle SEED=9904 CASES=30 EDGE=1 F64=both
run fuzz_kept SEED=9905 CASES=60
run fuzz_lbs SEED=9906 CASES=40
run fuzz_determinism SEED=9907 CASES=12
run fuzz_graphs SEED=9908 CASES=6
run fuzz_raster_closeup SEED=9909 CASES=12
cat gpurun_out/fuzz9/summary.txt
