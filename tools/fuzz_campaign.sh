mkdir -p gpurun_out/fuzz8
run() { name=$1; shift; ( timeout 600 env "$@" python tools/$name.py > gpurun_out/fuzz8/$name.$RANDOM.log 2>&1; echo "$name $* rc=$?" >> gpurun_out/fuzz8/summary.txt ); }
run fuzz_raster SEED=9801 CASES=120
run fuzz_raster_grads SEED=9802 CASES=100
run fuzz_cycle SEED=9803 CASES=40 F64=both
run fuzz_cycle SEED=9804 CASES=30 EDGE=1 F64=both
run fuzz_kept SEED=9805 CASES=60
run fuzz_lbs SEED=9806 CASES=40
run fuzz_determinism SEED=9807 CASES=12
run fuzz_graphs SEED=9808 CASES=6
run fuzz_raster_closeup SEED=9809 CASES=12
cat gpurun_out/fuzz8/summary.txt
