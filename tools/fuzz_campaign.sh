mkdir -p gpurun_out/fuzz7
run() { name=$1; shift; ( timeout 600 env "$@" python tools/$name.py > gpurun_out/fuzz7/$name.$RANDOM.log 2>&1; echo "$name $* rc=$?" >> gpurun_out/fuzz7/summary.txt ); }
run fuzz_raster SEED=9701 CASES=120
run fuzz_raster_grads SEED=9702 CASES=100
run fuzz_cycle SEED=9703 CASES=40 F64=both
run fuzz_cycle SEED=9704 CASES=30 EDGE=1 F64=both
run fuzz_kept SEED=9705 CASES=60
run fuzz_lbs SEED=9706 CASES=40
run fuzz_determinism SEED=9707 CASES=12
run fuzz_graphs SEED=9708 CASES=6
run fuzz_raster_closeup SEED=9709 CASES=12
cat gpurun_out/fuzz7/summary.txt
