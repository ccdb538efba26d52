#!/bin/bash
# developer tool (run on the GPU box through gpurun): the rocprofv3 evidence of one round, condensed under
# gpurun_out/profiles_<tag>/ -- copy what is to be judged into profiles/.
#   bash tools/profile_round.sh r02
TAG=${1:-rXX}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
W=$R/gpurun_out/prof_$TAG
rm -rf $W; mkdir -p $W $R/gpurun_out/profiles_$TAG
CMD="python $R/bench.py --no-cpu-baseline --no-fit --steps 20 --warmup 5"
rocprofv3 --kernel-trace --stats --output-format csv -d $W/stats -o s -- $CMD > $W/stats.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $W/fetch -o f -- $CMD > $W/fetch.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $W/write -o w -- $CMD > $W/write.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM -d $W/sq1 -o q -- $CMD > $W/sq1.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA -d $W/sq2 -o q -- $CMD > $W/sq2.log 2>&1
cd $R
python tools/make_profiles.py $TAG $W/stats $W/fetch $W/write gpurun_out/profiles_$TAG > /dev/null
mkdir -p $W/sq; cp -r $W/sq1 $W/sq2 $W/sq/
python tools/pmc_summary.py $W/sq > gpurun_out/profiles_$TAG/${TAG}_pmc_sq_counters.txt
rm -rf $W
ls -la gpurun_out/profiles_$TAG
