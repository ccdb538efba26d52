mkdir -p gpurun_out/fuzz4
run() { name=$1; shift; ( timeout 600 env "$@" python tools/$name.py > gpurun_out/fuzz4/$name.$RANDOM.log 2>&1; echo "$name $* rc=$?" >> gpurun_out/fuzz4/summary.txt ); }
run fuzz_raster SEED=9001 CASES=120
run fuzz_raster_grads SEED=9002 CASES=100
run fuzz_cycle SEED=9003 CASES=40
run fuzz_cycle SEED=9004 CASES=30 EDGE=1
run fuzz_kept SEED=9005 CASES=60
run fuzz_lbs SEED=9006 CASES=40
run fuzz_determinism SEED=9007 CASES=12
run fuzz_graphs SEED=9008 CASES=6
run fuzz_raster_closeup SEED=9009 CASES=12
cat gpurun_out/fuzz4/summary.txt
