#!/bin/bash
# Dev tool: build a variant of liblipmpc.so with extra compiler flags into variants/<name>.so (only the instantiations
# given in INSTS are rebuilt with the flags; the rest come from the normal build), for A/B timing on the GPU box:
#   tools/build_variant.sh nofresh "-DLIPMPC_NO_FRESH" "16_5"
# then on the box:  cp variants/nofresh.so humanoid-navigation-using-mpc-ldcbf_amd/liblipmpc.so && python tools/iter_cost.py
set -e
name=$1; flags=$2; insts=${3:-16_5}
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/humanoid-navigation-using-mpc-ldcbf_amd/csrc
mkdir -p $R/variants $C/build_$name
objs=""
for o in $C/build/*.o; do
  b=$(basename $o .o)
  use=$o
  for i in $insts; do
    if [ "$b" == "inst_$i" ]; then
      g=${i%_*}; n=${i#*_}
      hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -DINST_G=$g -DINST_NL=$n -c $C/lipmpc_inst.hip -o $C/build_$name/$b.o
      use=$C/build_$name/$b.o
    fi
  done
  objs="$objs $use"
done
hipcc --offload-arch=gfx950 -shared -fPIC -o $R/variants/$name.so $objs
echo built $R/variants/$name.so
