#!/bin/bash
# second GPU call of round 2: VMM interleave experiment, GPU test suite, in-LDS A/B, bench
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r02_b
mkdir -p $OUT
cd $R
timeout 400 tools/microbench/placement_study vmm 8 140 > $OUT/vmm.txt 2>&1
timeout 1200 python -m pytest tests -m gpu -q --timeout 300 -x > $OUT/pytest.txt 2>&1
timeout 600 python tools/ab_variants.py base=smfft_amd/libsmfft_amd.so read2=smfft_amd/libsmfft_amd_read2.so mw5=smfft_amd/libsmfft_amd_mw5.so --sizes 1024,256,4096,32 --paths multiple,rc --mult 1,10 > $OUT/ab_mult.txt 2>&1
timeout 300 python tools/ab_variants.py base=smfft_amd/libsmfft_amd.so read2=smfft_amd/libsmfft_amd_read2.so --sizes 1024,512,2048 --paths external > $OUT/ab_ext.txt 2>&1
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -5 $OUT/pytest.txt
