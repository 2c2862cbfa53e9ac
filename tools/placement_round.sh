#!/bin/bash
# Runs on the GPU box (via gpurun): the placement study of round 2 (tools/microbench/placement_study.hip):
# the chunk map with pure read / write rates, and PMC passes over tagged (input, output) class pairs.
# Usage: tools/placement_round.sh <tag>     -> everything under gpurun_out/<tag>/
set -u
TAG=${1:-r02_placement}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
B=$R/tools/microbench/placement_study
timeout 300 $B map 4 64 > $OUT/map.txt 2>&1
cd /tmp && export TMPDIR=/tmp
pass() {  # name, counters...
  local name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc_$name -- $B pmc 4 64 > $OUT/pmc_$name.txt 2>&1
}
pass ea_level TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum
pass ea_stall TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum
pass tcp_lat TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum
pass utcl TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_PENDING_STALL_CYCLES_sum GRBM_UTCL2_BUSY GRBM_EA_BUSY GRBM_GUI_ACTIVE
pass tcc TCC_BUSY_sum TCC_TAG_STALL_sum TCC_REQ_sum TCC_BUBBLE_sum
pass ta TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TD_TC_STALL_sum
ls -R $OUT | head -50
