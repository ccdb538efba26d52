mkdir -p gpurun_out/fuzz_r06
run() { name=$1; shift; ( timeout 600 env "$@" python tools/$name.py > gpurun_out/fuzz_r06/$name.$RANDOM.log 2>&1; echo "$name $* rc=$?" >> gpurun_out/fuzz_r06/summary.txt ); }
run fuzz_raster SEED=6601 CASES=120
run fuzz_raster_grads SEED=6602 CASES=100
run fuzz_cycle SEED=6603 CASES=40 F64=both
run fuzz_cycle SEED=6604 CASES=30 EDGE=1 F64=both
run fuzz_kept SEED=6605 CASES=60
run fuzz_lbs SEED=6606 CASES=40
run fuzz_determinism SEED=6607 CASES=12
run fuzz_graphs SEED=6608 CASES=6
run fuzz_raster_closeup SEED=6609 CASES=12
cat gpurun_out/fuzz_r06/summary.txt
