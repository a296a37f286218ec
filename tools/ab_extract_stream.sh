#!/bin/bash
# Same box, back to back: what the extract stream carries between two gathers (VERDICT r04 "a new counter that says
# where time goes": profiles/r05_ab_extract_stream.txt -- the trace shows 28 us of dead time between two gathers).
#   prev    : bench_prev.py, the tree before (git show 9781a65:bench.py): wait(sampled), record, GATHER, record, wait(labelled), record
#   events  : GGMS_BENCH_GATHER_TIMING=events: wait(sampled), record, GATHER, record   (labels waited for on the sampling stream)
#   timer   : the default: wait(sampled), GATHER -- timing and "rows are out" ride on the dispatch packet (ggms_launch_timer_t)
#   lean    : GGMS_BENCH_STEP_EVENTS=0: no per-step timing on either stream (a bound: what is left is the wait packet)
F="--no-engine --no-configs --no-staged-host --no-host-tier --no-cpu-baseline --no-sampler-roofline --pipelines 1 --extract-streams 1 $*"
pick='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print("%-7s" % sys.argv[1], "ms/step %.4f" % d["ms_per_step"], "edges/s %.4g" % d["value"], "gather us %.1f" % d["roofline"]["avg_launch_us"], "alone %.1f" % (d["roofline"]["avg_launch_us_alone"] or 0), "sample ms %.3f" % d["per_gpu"]["sample_ms_per_step"])'
for rep in 1 2 3; do
  [ -f bench_prev.py ] && python bench_prev.py $F 2>/dev/null | python -c "$pick" prev
  GGMS_BENCH_GATHER_TIMING=events python bench.py $F 2>/dev/null | python -c "$pick" events
  python bench.py $F 2>/dev/null | python -c "$pick" timer
  GGMS_BENCH_STEP_EVENTS=0 python bench.py $F 2>/dev/null | python -c "$pick" lean
done
