#!/bin/bash
# Experiment builds of the kernels with other compile-time choices (see the #ifndef blocks of csrc/ssw_kernels.hip):
#   scripts/build_variants.sh name "-DFLAG=.. -DFLAG=.." [name2 "..."] ...
# -> complete-striped-smith-waterman-library_amd/variants/libssw_<name>.so, selected at run time with SSW_LIB=<path>.
# The host objects are the ones of the normal build (run `make` first).
set -e
cd "$(dirname "$0")/../complete-striped-smith-waterman-library_amd"
mkdir -p variants build
while [ $# -ge 2 ]; do
	name=$1; flags=$2; shift 2
	/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../include -Icsrc $flags -c csrc/ssw_kernels.hip -o build/ssw_kernels_$name.o
	/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libssw_$name.so build/ssw_kernels_$name.o build/ssw_host.o build/ssw_pool.o build/ssw_cigar.o -lpthread
	echo "built variants/libssw_$name.so ($flags)"
done
