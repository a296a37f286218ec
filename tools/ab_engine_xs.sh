#!/bin/bash
# Same box: the engine (samgraph.torch surface, child process) with two extract streams (default) and with one
F="--no-configs --no-staged-host --no-host-tier --no-cpu-baseline --no-sampler-roofline --pipelines 1 --extract-streams 1 $*"
pick='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); e=d.get("engine",{}); print("%-8s" % sys.argv[1], "engine ms/step %s" % e.get("ms_per_step"), "edges/s %s" % e.get("edges_per_s"), "extract GB/s %s" % e.get("feature_GBps"), e.get("error",""))'
for rep in 1 2 3; do
  python bench.py $F 2>/dev/null | python -c "$pick" two
  SAMGRAPH_EXTRACT_STREAMS=1 python bench.py $F 2>/dev/null | python -c "$pick" one
done
