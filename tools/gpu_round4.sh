#!/bin/bash
# fourth GPU call of round 2: allocator (mixed policy) tests + bench, reads per site A/B, PMC of the in-LDS kernels
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r02_d
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -m gpu -q --timeout 300 -x -s -k "malloc_pair or bench_two or harness" > $OUT/pytest.txt 2>&1
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err
timeout 600 python tools/ab_variants.py r01=smfft_amd/libsmfft_amd_r01.so base=smfft_amd/libsmfft_amd.so --sizes 32,64,128,256,512,1024,2048,4096 --paths multiple,external,rc --mult 1 > $OUT/ab_all.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $OUT/pmc_lds -- python3 $R/tools/sweep.py --sizes 32,64,128,256,512,1024,2048,4096 --paths multiple --variants f0,f1 --rounds 2 > $OUT/pmc_lds.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/pmc_wait -- python3 $R/tools/sweep.py --sizes 256,1024,4096 --paths multiple --variants f0,f1 --rounds 2 > $OUT/pmc_wait.txt 2>&1
cd $R
tail -5 $OUT/pytest.txt
