#!/bin/bash
# Runs on the GPU box (via gpurun): bench line, rocprofv3 kernel stats of the same command, PMC
# passes (HBM traffic; LDS bank conflicts on the in-LDS path), and the config 2/3/4 runner.
# Usage: tools/profile_round.sh <tag>     -> everything under gpurun_out/<tag>/
set -u
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
python bench.py > $OUT/bench.json 2> $OUT/bench.err
cp bench_detail.json $OUT/bench_detail.json     # the full record behind the compact line (later runs overwrite bench_detail.json)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-plain > $OUT/bench_under_rocprof.json 2>/dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $OUT/pmc_lds -- python3 $R/tools/sweep.py --sizes 32,64,128,256,512,1024,2048,4096 --paths multiple,external --variants f0,f1 --rounds 2 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc_wait -- python3 $R/tools/sweep.py --sizes 32,64,128,256,512,1024,2048,4096 --paths multiple --variants f0,f1 --rounds 2 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_calib_fetch -- $R/tools/microbench/membw 268435456 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_calib_write -- $R/tools/microbench/membw 268435456 > /dev/null 2>&1
ls $OUT
