#!/bin/bash
# Occupancy of the wave-per-task factor launches at ranks 17 .. 32 (esplit_latent<20 / 24 / 32, false>): variants of the
# library with other __launch_bounds__ minimum-waves values, built ON THE GPU BOX into /tmp, each timed on the first three EM
# iterations of a cold fit (ranks 29 / 29 / 21).  gpurun -- 'bash tools/estep_lb_sweep.sh "1,1 3,5 4,5 4,6"'
set -u
out=gpurun_out/lb; mkdir -p $out
b=/tmp/lbbuild; rm -rf $b; mkdir -p $b/vlgp_amd $b/include
cp -r vlgp_amd/csrc $b/vlgp_amd/; cp include/*.h $b/include/
cd $b/vlgp_amd/csrc
for v in ${1:-"1,1 3,5 4,5"}; do
  lb32=${v%,*}; lb24=${v#*,}
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DESPLIT_LB32=$lb32 -DESPLIT_LB24=$lb24 -Wno-unused-function -Wno-unused-variable -c estep_split.hip -o estep_split.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC api.o estep.o estep_fast.o estep_split.o estep_long.o mstep.o hstep.o prior.o misc.o sample.o -o $b/lib_${lb32}_${lb24}.so -ldl -lrt || exit 1
  for rep in 1 2; do
    ( cd $OLDPWD && VLGP_LIB_PATH=$b/lib_${lb32}_${lb24}.so STEPS=4 python tools/estep_per_step.py > $out/lb${lb32}_${lb24}_$rep.txt 2>&1 )
    echo "LB32=$lb32 LB24=$lb24 rep $rep: $(grep -E '^it +[123] ' $OLDPWD/$out/lb${lb32}_${lb24}_$rep.txt | sed 's/ranks.*\]  //' | tr '\n' '|')"
  done
done
