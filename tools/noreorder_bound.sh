#!/bin/bash
# What the bit-reversed re-read costs the planar no-reorder kernels: a TIMING-ONLY build (wrong results) whose
# no-reorder loop forwards an application's registers into the next one as the natural-order kernel does, next to the product, in one
# process on the same buffers.  Run on the GPU box:  tools/noreorder_bound.sh  ->  gpurun_out/r06_noreorder_bound.txt
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
mkdir -p build_ab gpurun_out
tools/build_variant.sh timing_fwd -DSMFFT_TIMING_ONLY_NOREORDER_FORWARD=1
mv smfft_amd/libsmfft_amd_timing_fwd.so build_ab/
rm -rf smfft_amd/csrc/build_timing_fwd
python tools/ab_variants.py product=smfft_amd/libsmfft_amd.so timing_only_forward=build_ab/libsmfft_amd_timing_fwd.so --sizes 128,256,512,1024,2048,4096 --paths multiple --plain --rounds 9 | tee gpurun_out/r06_noreorder_bound.txt
