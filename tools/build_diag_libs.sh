#!/bin/bash
# Variant builds of libdprnn_mi355x.so that differ only in ONE object (recur_split8_f16.o, or SRC=<name> without .hip): tools/build_diag_libs.sh OUTDIR "name:flags" ...
#   e.g. tools/build_diag_libs.sh tools/probes/diag "diag4:-DDPRNN_SPLIT8_DIAG=4" "sep:-DDPRNN_CELL_SHARED_RCP=0"
# (the other objects are the tree's own: run `make -C tss-with-dprnn_amd/csrc` first).  *.so is git-ignored and travels with gpurun.
set -e
out=$1; shift
src=tss-with-dprnn_amd/csrc
unit=${SRC:-recur_split8_f16}
mkdir -p "$out"
make -s -j8 -C $src
pids=()
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  ( /opt/rocm/bin/hipcc $flags -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -Wall -Wno-unused-function -c $src/$unit.hip -o $out/$name.o &&
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $(ls $src/*.o | grep -v "/$unit.o") $out/$name.o -o $out/libdprnn_$name.so && rm $out/$name.o && echo "built $out/libdprnn_$name.so ($flags)" ) &
  pids+=($!)
  if [ ${#pids[@]} -ge 4 ]; then wait ${pids[0]}; pids=("${pids[@]:1}"); fi
done
wait
