set -o pipefail
pair() {
  timeout -k 10 600 python tools/soak_batches.py 40 $1 > /tmp/soak_a.txt 2>&1 &
  A=$!
  timeout -k 10 600 python tools/soak_batches.py 40 $2 > /tmp/soak_b.txt 2>&1 &
  B=$!
  wait $A; ra=$?; wait $B; rb=$?
  echo "== $1 beside $2: rc $ra $rb"; grep soak /tmp/soak_a.txt /tmp/soak_b.txt | sed 's/^.tmp.soak_//'
  [ $ra -eq 0 ] && [ $rb -eq 0 ]
}
pair khop0 khop1 && pair khop2 khop0 && pair khop1 random_walk
