#!/bin/bash
# Build a copy of libraytracing_amd.so with extra kernel compile flags (development aid):
#   tools/build_variant.sh NAME -DRT_GROUP_LDS=2 ...   ->  variants/libNAME.so
# Variants are built with -DRT_DEV_KNOBS: they read the RT_DEBUG_* environment variables, the shipped library does not.
# Run with RT_AMD_LIB=variants/libNAME.so.
set -e
cd "$(dirname "$0")/../raytracing_amd/csrc"
name=$1; shift
set -- -DRT_DEV_KNOBS "$@"
mkdir -p ../../variants build
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I../../include -I. \
    --offload-arch=gfx950 "$@" -c rt_kernels.hip -o ../../variants/rt_kernels_$name.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I../../include -I. \
    "$@" -c rt_api.cpp -o ../../variants/rt_api_$name.o
objs=$(ls build/*.o | grep -v "rt_kernels.o\|rt_api.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../variants/lib$name.so ../../variants/rt_kernels_$name.o \
    ../../variants/rt_api_$name.o $objs
rm ../../variants/rt_kernels_$name.o ../../variants/rt_api_$name.o
echo variants/lib$name.so
