#!/bin/bash
# dev tool: run one test against alternative builds of the library (liblipmpc_<variant>.so next to liblipmpc.so)
cd "$(dirname "$0")/.."
P=humanoid-navigation-using-mpc-ldcbf_amd
cp $P/liblipmpc.so /tmp/liblipmpc_orig.so
for v in "$@"; do
  cp $P/liblipmpc_$v.so $P/liblipmpc.so
  echo "== variant $v"
  timeout -k 10 200 python -m pytest tests -m gpu -x -q -k "config4 or two_row" 2>&1 | grep -E "cfg4 statuses|passed|failed"
done
cp /tmp/liblipmpc_orig.so $P/liblipmpc.so
