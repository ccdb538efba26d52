mkdir -p gpurun_out/fuzz6
run() { name=$1; shift; ( timeout 600 env "$@" python tools/$name.py > gpurun_out/fuzz6/$name.$RANDOM.log 2>&1; echo "$name $* rc=$?" >> gpurun_out/fuzz6/summary.txt ); }
run fuzz_raster SEED=9201 CASES=120
run fuzz_raster_grads SEED=9202 CASES=100
run fuzz_cycle SEED=9203 CASES=40 F64=both
run fuzz_cycle SEED=9204 CASES=30 EDGE=1 F64=both
run fuzz_kept SEED=9205 CASES=60
run fuzz_lbs SEED=9206 CASES=40
run fuzz_determinism SEED=9207 CASES=12
run fuzz_graphs SEED=9208 CASES=6
run fuzz_raster_closeup SEED=9209 CASES=12
cat gpurun_out/fuzz6/summary.txt
