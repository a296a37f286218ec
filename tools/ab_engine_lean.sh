#!/bin/bash
# Same box: the engine (samgraph.torch surface, child process) with the lean extract stream (default) and with the
# sequence before it (SAMGRAPH_LEAN_EXTRACT=0); the bench's own step beside it.
F="--no-configs --no-staged-host --no-host-tier --no-cpu-baseline --no-sampler-roofline $*"
pick='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); e=d.get("engine",{}); print("%-8s" % sys.argv[1], "bench ms/step %.4f" % d["ms_per_step"], "| engine ms/step %s" % e.get("ms_per_step"), "edges/s %s" % e.get("edges_per_s"), "sample %s extract GB/s %s" % (e.get("sample_edges_per_s"), e.get("feature_GBps")), e.get("error",""))'
for rep in 1 2 3; do
  python bench.py $F 2>/dev/null | python -c "$pick" lean
  SAMGRAPH_LEAN_EXTRACT=0 python bench.py $F 2>/dev/null | python -c "$pick" before
done
