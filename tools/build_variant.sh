#!/bin/bash
# Builds a variant of the library next to the product for tools/ab_variants.py:
#   tools/build_variant.sh <tag> [hipcc flags...]   ->  smfft_amd/libsmfft_amd_<tag>.so
set -e
TAG=$1; shift
cd "$(dirname "$0")/../smfft_amd/csrc"
make -j8 LIB=../libsmfft_amd_$TAG.so OBJDIR=build_$TAG EXTRA_HIPFLAGS="$*" ../libsmfft_amd_$TAG.so 2>&1 | grep -E "error|warning" || true
ls -la ../libsmfft_amd_$TAG.so
