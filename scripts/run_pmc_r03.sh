#!/bin/bash
# rocprofv3 counter passes of round 3 (run through gpurun from the repo root):  bash scripts/run_pmc_r03.sh <outdir> [sets]
# sets: any of  ro (k_rollout)  mc (search tick)  tw (towers: matrix-core counters)   -- default all
# separate passes as MI355X_MICROARCH.md prescribes (FETCH_SIZE and WRITE_SIZE do not fit one pass); the program itself
# directly after `--`
set -o pipefail
OUT=${1:-gpurun_out/r03_pmc}
SETS=${2:-"ro mc tw"}
R=$(pwd)
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
pass() {  # name, script, counters...
  local name=$1 script=$2; shift 2
  rocprofv3 --pmc "$@" --kernel-trace -d $R/$OUT/$name -o $name --output-format csv -- python3 $R/scripts/$script > $R/$OUT/$name.log 2>&1 || { echo "pass $name failed"; tail -5 $R/$OUT/$name.log; return 1; }
  echo "pass $name done"
}
for s in $SETS; do
  case $s in
  ro)
    pass ro_inst rollout_pmc.py SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES &&
    pass ro_busy rollout_pmc.py GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES &&
    pass ro_wait rollout_pmc.py SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES &&
    pass ro_fetch rollout_pmc.py FETCH_SIZE &&
    pass ro_write rollout_pmc.py WRITE_SIZE || exit 1 ;;
  mc)
    pass mc_inst mcts_ticks_only.py SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES &&
    pass mc_wait mcts_ticks_only.py SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES &&
    pass mc_busy mcts_ticks_only.py GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR &&
    pass mc_fetch mcts_ticks_only.py FETCH_SIZE &&
    pass mc_write mcts_ticks_only.py WRITE_SIZE &&
    pass mc_tcc mcts_ticks_only.py TCC_HIT_sum TCC_MISS_sum || exit 1 ;;
  tw)
    # matrix-core utilisation of the split towers: SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CYCLES ...) -- see
    # scripts/pmc_mfma_util.py for the arithmetic; LDS bank conflicts beside it
    pass tw_mfma towers_only.py SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES &&
    pass tw_lds towers_only.py SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES &&
    pass tw_busy towers_only.py GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY || exit 1 ;;
  esac
done
cd $R
for p in ro_inst ro_busy ro_wait ro_fetch ro_write; do [ -d $OUT/$p ] && { echo "== $p"; python3 scripts/pmc_summary.py $OUT/$p k_rollout --per-dispatch; }; done > $OUT/rollout_pmc.txt 2>&1
for p in mc_inst mc_wait mc_busy mc_fetch mc_write mc_tcc; do [ -d $OUT/$p ] && { echo "== $p"; python3 scripts/pmc_summary.py $OUT/$p; }; done > $OUT/mcts_pmc.txt 2>&1
for p in tw_mfma tw_lds tw_busy; do [ -d $OUT/$p ] && { echo "== $p"; python3 scripts/pmc_summary.py $OUT/$p k_trunk; }; done > $OUT/tower_pmc.txt 2>&1
# keep the merge small: the raw csv of the search passes is 900 dispatches x counters
find $OUT -name "*.csv" -size +2M -delete
echo finished
