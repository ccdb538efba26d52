mkdir -p gpurun_out/fuzz_r05
run() { name=$1; shift; ( timeout 600 env "$@" python tools/$name.py > gpurun_out/fuzz_r05/$name.$RANDOM.log 2>&1; echo "$name $* rc=$?" >> gpurun_out/fuzz_r05/summary.txt ); }
run fuzz_raster SEED=5501 CASES=120
run fuzz_raster_grads SEED=5502 CASES=100
run fuzz_cycle SEED=5503 CASES=40 F64=both
run fuzz_cycle SEED=5504 CASES=30 EDGE=1 F64=both
run fuzz_kept SEED=5505 CASES=60
run fuzz_lbs SEED=5506 CASES=40
run fuzz_determinism SEED=5507 CASES=12
run fuzz_graphs SEED=5508 CASES=6
run fuzz_raster_closeup SEED=5509 CASES=12
cat gpurun_out/fuzz_r05/summary.txt
