#!/bin/bash
# dev tool: run the config-4 test and bench against alternative builds (liblipmpc_<variant>.so next to liblipmpc.so)
cd "$(dirname "$0")/.."
P=humanoid-navigation-using-mpc-ldcbf_amd
cp $P/liblipmpc.so /tmp/liblipmpc_orig.so
for v in "$@"; do
  cp $P/liblipmpc_$v.so $P/liblipmpc.so
  echo "== variant $v"
  timeout -k 10 200 python -m pytest tests -m gpu -x -q -k config4 2>&1 | grep -E "cfg4 statuses gpu \[|passed|failed"
  timeout -k 10 200 python bench.py --horizon 16 --obstacles 50 --batch 4096 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | cut -c60-140
done
cp /tmp/liblipmpc_orig.so $P/liblipmpc.so
