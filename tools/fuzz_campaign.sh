mkdir -p gpurun_out/fuzz_r06b
run() { name=$1; shift; ( timeout 600 env "$@" python tools/$name.py > gpurun_out/fuzz_r06b/$name.$RANDOM.log 2>&1; echo "$name $* rc=$?" >> gpurun_out/fuzz_r06b/summary.txt ); }
run fuzz_raster SEED=6701 CASES=120
run fuzz_raster_grads SEED=6702 CASES=100
run fuzz_cycle SEED=6703 CASES=40 F64=both
run fuzz_cycle SEED=6704 CASES=30 EDGE=1 F64=both
run fuzz_kept SEED=6705 CASES=60
run fuzz_lbs SEED=6706 CASES=40
run fuzz_determinism SEED=6707 CASES=12
run fuzz_graphs SEED=6708 CASES=6
run fuzz_raster_closeup SEED=6709 CASES=12
cat gpurun_out/fuzz_r06b/summary.txt
