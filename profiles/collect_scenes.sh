#!/bin/bash
# Per-scene counter summaries behind bench.py's `traffic` / VALU fields (round 4): bash profiles/collect_scenes.sh r04 [scenes]
# For every scene of bench.py (untrained, trained, densified) K steps of profiles/scene_step.py run between two marker kernels
# under separate rocprofv3 --pmc passes (counters only with --kernel-trace): FETCH_SIZE, WRITE_SIZE, two SQ sets; plus one
# --kernel-trace --stats pass.  The summaries SUM every kernel's launches inside the window and divide by K.
#   -> gpurun_out/prof_<round>/{pmc_hbm_traffic,sq_counters,kernel_stats}_<scene>.csv   (copy into profiles/<round>/)
set -e
R=${1:-r04}
SCENES=${2:-"untrained trained densified"}
K=6
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for S in $SCENES; do
  MF=/tmp/w3d_scene_$S.pt
  rm -f $MF
  python3 profiles/scene_step.py --scene $S --steps 2 --model-file $MF > $OUT/prep_$S.log 2>&1      # prepare + save the model
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks_$S -o s -- python3 profiles/scene_step.py --scene $S --steps 30 --model-file $MF > $OUT/ks_$S.log 2>&1
  cp $OUT/ks_$S/s_kernel_stats.csv $OUT/kernel_stats_$S.csv
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_${S}_$C -o p -- python3 profiles/scene_step.py --scene $S --steps $K --model-file $MF > $OUT/pmc_${S}_$C.log 2>&1
  done
  python3 profiles/summarize_pmc.py --window lgamma --steps $K $OUT/pmc_${S}_FETCH_SIZE/p_counter_collection.csv $OUT/pmc_${S}_WRITE_SIZE/p_counter_collection.csv > $OUT/pmc_hbm_traffic_$S.csv
  i=0
  for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY"; do
    i=$((i+1))
    rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/sq_${S}_$i -o p -- python3 profiles/scene_step.py --scene $S --steps $K --model-file $MF > $OUT/sq_${S}_$i.log 2>&1
  done
  python3 profiles/summarize_sq.py --window lgamma --steps $K $OUT/sq_${S}_1/p_counter_collection.csv $OUT/sq_${S}_2/p_counter_collection.csv > $OUT/sq_counters_$S.csv
  rm -rf $OUT/ks_$S $OUT/pmc_${S}_FETCH_SIZE $OUT/pmc_${S}_WRITE_SIZE $OUT/sq_${S}_1 $OUT/sq_${S}_2
  echo "== $S"; head -14 $OUT/pmc_hbm_traffic_$S.csv; head -6 $OUT/sq_counters_$S.csv | cut -c1-160
done
