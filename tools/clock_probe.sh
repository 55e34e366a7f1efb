#!/bin/bash
# sample shader clock / power with rocm-smi while a conv configuration runs in a loop.  $1 = shape, $2 = cfg
python3 tools/conv_bench.py 8 400 $1 $2 > /tmp/cb.log 2>&1 &
PID=$!
sleep 6
for i in 1 2 3 4 5 6; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo
  sleep 0.7
done
wait $PID
tail -1 /tmp/cb.log
