for st in weighted_khop_hash_dedup khop0 khop2 khop1 weighted_khop weighted_khop_prefix; do
  echo "== products $st"; bash tools/trace_kernels.sh r03f_$st --preset products --sample-type $st --no-overlap --steps 10 2>&1 | cut -c1-150 | grep -v "rocclr\|init_states"
done
echo "== friendster random_walk"; bash tools/trace_kernels.sh r03f_rw --preset friendster --sample-type random_walk --fanout 5,5,5 --no-overlap --steps 10 2>&1 | cut -c1-150 | grep -v "rocclr\|init_states"
