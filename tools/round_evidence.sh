#!/bin/bash
# developer tool (GPU box, through gpurun): the measured evidence of a round beyond the rocprofv3 passes of profile_round.sh --
# the gfx950 issue-cost micro-benchmarks and the phase timings of the four large kernels (timing builds under variants/:
# tools/mkvariant.sh t1..t5 mh_raster.hip -DMH_EXPERIMENT -DR_TIMING=1..5, lt mh_lbs.hip -DMH_EXPERIMENT -DLBS_TIMING) -- into gpurun_out/evidence_<tag>/
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/evidence_$TAG
mkdir -p $O
cd $R
export PYTHONPATH=$R/scene-aware-3d-multi-human_amd
{
  echo "# tools/ubench/valu_rate.hip: issue cost of VALU instruction classes, gfx950 (ns per wave64 instruction per SIMD at 1 and 4 waves/SIMD)"
  tools/ubench/valu_rate.bin
  echo
  echo "# tools/ubench/lds_rate.hip: LDS instruction cost by instruction, active lanes, address stride (16 waves per CU issuing; wall ms x 23.4 = cycles per wave-instruction per CU)"
  tools/ubench/lds_rate.bin
} > $O/${TAG}_ubench_valu_lds.txt 2>&1
{
  echo "# wave-elapsed shader cycles by phase, summed over the waves of ONE C3 launch (tools/pair_stats.py with the timing builds)"
  echo "# k_raster_strip (builds 1, 2): [tile prologue, round head, cull walk + pair list, pair evaluation], [even-split path, wait for the tile's other waves, tile epilogue, everything]; build 3: workgroup life spans vs the kernel's span"
  for t in 1 2 3; do MHHIP_LIB=$R/variants/lib_t$t.so R_TIMING=$t python tools/pair_stats.py 2>&1 | grep "timing build"; done
  echo "# k_raster_grads (builds 4, 5): [unit header + body sums, classification loads + table clear, compaction, pixels], [reductions + flush, -, -, everything]"
  for t in 4 5; do MHHIP_LIB=$R/variants/lib_t$t.so R_TIMING=$t python tools/pair_stats.py 2>&1 | grep "timing build"; done
  echo "# k_skin_fwd16 / k_skinbwd16 inside the replayed cycle (tools/time_lbs_phases.py, -DMH_EXPERIMENT -DLBS_TIMING)"
  MHHIP_LIB=$R/variants/lib_lt.so python tools/time_lbs_phases.py 2>&1 | grep "per wave"
} > $O/${TAG}_phase_timings.txt 2>&1
ls -la $O
