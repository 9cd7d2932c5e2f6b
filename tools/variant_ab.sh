#!/bin/bash
# Same-box A/B of compile-time variants of ONE source of the library, built ON THE GPU BOX into /tmp:
#   gpurun -- 'CMD="python tools/estep_rank_classes.py" bash tools/variant_ab.sh estep_split.hip "" "-DLANE_LB14=1"'
# every variant (a string of -D flags; "" = the shipped build) is linked against the shipped objects of the other sources
# and $CMD runs twice with VLGP_LIB_PATH pointing at it; the outputs go to gpurun_out/ab/.
set -u
src=$1; shift
out=$PWD/gpurun_out/ab; mkdir -p $out
root=$PWD
b=/tmp/abbuild; rm -rf $b; mkdir -p $b/vlgp_amd $b/include
cp -r vlgp_amd/csrc $b/vlgp_amd/; cp include/*.h $b/include/
cd $b/vlgp_amd/csrc
i=0
for flags in "$@"; do
  i=$((i+1))
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $flags -Wno-unused-function -Wno-unused-variable -c $src -o ${src%.hip}.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC api.o estep.o estep_fast.o estep_split.o estep_long.o mstep.o hstep.o prior.o misc.o sample.o -o $b/lib_v$i.so -ldl -lrt || exit 1
  for rep in 1 2; do
    ( cd $root && VLGP_LIB_PATH=$b/lib_v$i.so bash -c "$CMD" > $out/v${i}_$rep.txt 2>&1 )
    echo "== variant $i [$flags] rep $rep"; grep -v "^$" $out/v${i}_$rep.txt | tail -${TAIL:-8}
  done
done
