#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/pmc_contract
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/wait -- python3 $R/tools/reference_contract.py --sizes 256,1024,4096 --in-lds-only > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAVES --output-format csv -d $OUT/lds -- python3 $R/tools/reference_contract.py --sizes 256,1024,4096 --in-lds-only > /dev/null 2>&1
cd $R
python tools/pmc_report.py $OUT/wait > $OUT/wait.txt 2>&1
python tools/pmc_report.py $OUT/lds > $OUT/lds.txt 2>&1
grep -i "multiple" $OUT/wait.txt | head -20
grep -i "multiple" $OUT/lds.txt | head -20
