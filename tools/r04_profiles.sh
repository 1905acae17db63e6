#!/bin/bash
# The rocprofv3 evidence of the round, on a GPU box: kernel-trace statistics of the bench command on both indexes and of the
# seed lookup on synthetic 2 x 2 GiB / 2 x 8 GiB tables, and PMC passes (FETCH_SIZE and WRITE_SIZE need a pass each; SQ group).
# Writes gpurun_out/r04_*; the summaries are copied into profiles/ by hand.   tools/r04_profiles.sh
cd "$(dirname "$0")/.."
SQ="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE TA_TA_BUSY_sum"
MIX="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_FLAT SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_LDS"
PROF_HEADLINE=strain bash tools/prof_stats.sh r04_strain 65536 > /dev/null 2>&1; echo "kernel stats strain done"
PROF_HEADLINE=demo bash tools/prof_stats.sh r04_demo 65536 > /dev/null 2>&1; echo "kernel stats demo done"
PROF_SEED_HBM_MIB=2048 bash tools/prof_stats.sh r04_seed_hbm_2g 65536 > /dev/null 2>&1; echo "kernel stats seed 2g done"
PROF_SEED_HBM_MIB=8192 bash tools/prof_stats.sh r04_seed_hbm_8g 65536 > /dev/null 2>&1; echo "kernel stats seed 8g done"
PMC_HEADLINE=strain bash tools/pmc_run.sh r04_strain 65536 "FETCH_SIZE" "WRITE_SIZE" "$SQ" "$MIX" 2>&1 | tail -2
PMC_HEADLINE=demo bash tools/pmc_run.sh r04_demo 65536 "FETCH_SIZE" "WRITE_SIZE" "$SQ" 2>&1 | tail -2
PMC_SEED_HBM_MIB=2048 bash tools/pmc_run.sh r04_seed_hbm_2g 65536 "FETCH_SIZE" "WRITE_SIZE" "$SQ" 2>&1 | tail -1
PMC_SEED_HBM_MIB=8192 bash tools/pmc_run.sh r04_seed_hbm_8g 65536 "FETCH_SIZE" "WRITE_SIZE" 2>&1 | tail -1
ls gpurun_out | grep r04_
