#!/bin/bash
# Dev tool: build a variant of liblipmpc.so with extra compiler flags into variants/<name>.so (only the objects given in INSTS
# are rebuilt with the flags -- "16_5" = inst_16_5.o, "8:16_7" = the 8-variable inst8_16_7.o, "L:32_2" = the split-launch body
# list_32_2.o, "api" = the C ABI object, "lidar" = the LiDAR front end; the rest come from the normal build), for A/B timing on the GPU box through LIPMPC_LIB
# (the shipped library is never replaced):
#   tools/build_variant.sh phase "-DLIPMPC_PHASE_TIMING" "api 16_5 32_25 L:32_1 L:32_2 L:32_4"
#   LIPMPC_LIB=$PWD/variants/phase.so LIPMPC_ALLOW_VARIANT=1 python tools/phase_cycles.py
# Every dev switch of the kernel sources is listed in tools/variants.txt; `tools/build_variant.sh --check-all` compiles each
# of them (CPU box, no GPU needed), which is what tests/test_abi_and_c_oracle.py::test_dev_variants_compile runs.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/humanoid-navigation-using-mpc-ldcbf_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC"
if [ "$1" == "--check-all" ]; then
  # one representative object per switch: compile only (-c to /dev/null)
  while read -r name flags insts; do
    [[ -z "$name" || "$name" == \#* ]] && continue
    first=${insts%% *}
    if [ "$first" == "api" ]; then
      hipcc $FLAGS $flags -c $C/lipmpc_api.hip -o /dev/null
    elif [ "$first" == "lidar" ]; then
      hipcc $FLAGS $flags -c $C/lipmpc_lidar.hip -o /dev/null
    else
      key=${first#*:}; g=${key%_*}; n=${key#*_}; extra=""; nv=$g
      [[ $first == L:* ]] && extra="-DINST_LIST"
      [[ $first == 8:* ]] && nv=8
      hipcc $FLAGS $flags $extra -DINST_G=$g -DINST_NL=$n -DINST_NV=$nv -c $C/lipmpc_inst.hip -o /dev/null
    fi
    echo "variant $name ($flags): $first compiles"
  done < $R/tools/variants.txt
  exit 0
fi
name=$1; flags=$2; insts=${3:-16_5}
mkdir -p $R/variants $C/build_$name
objs=""
for o in $C/build/*.o; do
  b=$(basename $o .o)
  use=$o
  for i in $insts; do
    if [ "$i" == "api" ] || [ "$i" == "lidar" ]; then
      if [ "$b" == "$i" ]; then
        hipcc $FLAGS $flags -c $C/lipmpc_$i.hip -o $C/build_$name/$i.o
        use=$C/build_$name/$i.o
      fi
      continue
    fi
    nv=""; extra=""; key=${i#*:}
    if [[ $i == 8:* ]]; then nv=8; want="inst8_$key"; elif [[ $i == L:* ]]; then extra="-DINST_LIST"; want="list_$key"; else want="inst_$key"; fi
    if [ "$b" == "$want" ]; then
      g=${key%_*}; n=${key#*_}
      hipcc $FLAGS $flags $extra -DINST_G=$g -DINST_NL=$n -DINST_NV=${nv:-$g} -c $C/lipmpc_inst.hip -o $C/build_$name/$b.o
      use=$C/build_$name/$b.o
    fi
  done
  objs="$objs $use"
done
hipcc --offload-arch=gfx950 -shared -fPIC -o $R/variants/$name.so $objs
echo built $R/variants/$name.so
