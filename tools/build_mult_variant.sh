#!/bin/bash
# usage: build_mult_variant.sh <tag> <flags...>  -> build_ab/libsmfft_amd_<tag>.so : the product with its in-LDS objects recompiled with <flags> (N >= 128) 
set -e
TAG=$1; shift
cd /root/repo/smfft_amd/csrc
mkdir -p /tmp/obj_$TAG
for n in 128 256 512 1024 2048 4096; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -fno-slp-vectorize -Wno-unused-function -I../../include "$@" -DSMFFT_N=$n -DSMFFT_INST_PART=2 -c smfft_inst.hip -o /tmp/obj_$TAG/smfft_mult_$n.o &
done
wait
OBJS=""
for n in 32 64 128 256 512 1024 2048 4096; do OBJS="$OBJS build/smfft_inst_$n.o"; done
for n in 32 64; do OBJS="$OBJS build/smfft_mult_$n.o"; done
for n in 128 256 512 1024 2048 4096; do OBJS="$OBJS /tmp/obj_$TAG/smfft_mult_$n.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build_ab/libsmfft_amd_$TAG.so $OBJS build/smfft_api.o build/smfft_pairs.o build/smfft_stream.o -lpthread
ls -la ../../build_ab/libsmfft_amd_$TAG.so
