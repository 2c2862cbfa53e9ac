#!/bin/bash
# third GPU call of round 2: VMM assembly of mixed chunks, in-LDS A/B against the round-1 library, R2C prefetch A/B, new allocator + bench
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r02_c
mkdir -p $OUT
cd $R
timeout 300 tools/microbench/placement_study vmm 8 100 > $OUT/vmm8.txt 2>&1
timeout 300 tools/microbench/placement_study vmm 256 100 > $OUT/vmm256.txt 2>&1
timeout 900 python tools/ab_variants.py r01=smfft_amd/libsmfft_amd_r01.so base=smfft_amd/libsmfft_amd.so read2=smfft_amd/libsmfft_amd_read2.so --sizes 32,64,128,256,512,1024,2048,4096 --paths multiple --mult 1,10 > $OUT/ab_mult.txt 2>&1
timeout 300 python tools/ab_variants.py r01=smfft_amd/libsmfft_amd_r01.so base=smfft_amd/libsmfft_amd.so rcpf=smfft_amd/libsmfft_amd_rcpf.so --sizes 256,512,1024 --paths rc > $OUT/ab_rc.txt 2>&1
timeout 600 python -m pytest tests -m gpu -q --timeout 300 -x -k "malloc_pair or bench_two or harness or host_transform_families" > $OUT/pytest.txt 2>&1
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -5 $OUT/pytest.txt
