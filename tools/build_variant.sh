#!/bin/bash
# Dev tool: build a variant of liblipmpc.so with extra compiler flags into variants/<name>.so (only the instantiations
# given in INSTS are rebuilt with the flags -- "16_5" = inst_16_5.o, "8:16_7" = the 8-variable inst8_16_7.o; the rest come
# from the normal build), for A/B timing on the GPU box through LIPMPC_LIB (the shipped library is never replaced):
#   tools/build_variant.sh phase "-DLIPMPC_PHASE_TIMING" "16_5 32_25 32_0"
#   LIPMPC_LIB=$PWD/variants/phase.so python tools/phase_cycles.py
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
    nv=""; key=$i
    if [[ $i == 8:* ]]; then nv=8; key=${i#8:}; want="inst8_$key"; else want="inst_$key"; fi
    if [ "$b" == "$want" ]; then
      g=${key%_*}; n=${key#*_}
      hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -DINST_G=$g -DINST_NL=$n -DINST_NV=${nv:-$g} -c $C/lipmpc_inst.hip -o $C/build_$name/$b.o
      use=$C/build_$name/$b.o
    fi
  done
  objs="$objs $use"
done
hipcc --offload-arch=gfx950 -shared -fPIC -o $R/variants/$name.so $objs
echo built $R/variants/$name.so
