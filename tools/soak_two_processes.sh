#!/bin/bash
# Two soak processes sharing ONE GPU (each: two pipelines + an HBM stream, every batch compared with the oracle): the
# look-back chains of two processes compete for the device's workgroup slots.  Then the same soak alone with look-backs
# that never wait (patience 0).   usage: bash tools/soak_two_processes.sh
set -o pipefail
echo "== two processes on one GPU, default patience"
timeout -k 10 600 python tools/soak_batches.py 60 khop3 > /tmp/soak_a.txt 2>&1 &
A=$!
timeout -k 10 600 python tools/soak_batches.py 60 khop3 > /tmp/soak_b.txt 2>&1 &
B=$!
wait $A; ra=$?
wait $B; rb=$?
echo "process A rc $ra"; grep "soak" /tmp/soak_a.txt
echo "process B rc $rb"; grep "soak" /tmp/soak_b.txt
echo "== khop3 beside random_walk"
timeout -k 10 600 python tools/soak_batches.py 40 random_walk > /tmp/soak_a.txt 2>&1 &
A=$!
timeout -k 10 600 python tools/soak_batches.py 60 khop3 > /tmp/soak_b.txt 2>&1 &
B=$!
wait $A; ra2=$?
wait $B; rb2=$?
echo "process A rc $ra2"; grep "soak" /tmp/soak_a.txt
echo "process B rc $rb2"; grep "soak" /tmp/soak_b.txt
echo "== one process, patience 0"
rc0=0
for st in khop3 random_walk khop0 khop1; do
  GGMS_TEST_SCAN_PATIENCE=0 timeout -k 10 600 python tools/soak_batches.py 30 $st 2>&1 | grep "soak" || rc0=1
done
[ $ra -eq 0 ] && [ $rb -eq 0 ] && [ $ra2 -eq 0 ] && [ $rb2 -eq 0 ] && [ $rc0 -eq 0 ]
