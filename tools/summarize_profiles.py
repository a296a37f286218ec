"""Turn the raw rocprofv3 output of tools/collect_profiles.sh into the summaries kept under profiles/.

    python tools/summarize_profiles.py gpurun_out/profiles_<tag> profiles/<round>   [steps of the main region, default 63]

writes
    <round>_kernel_stats.csv        rocprofv3's own --stats table, ggms kernels + copies only
    <round>_kernel_timeline.csv     per kernel: launches per step, avg/min/max us, and per HSA queue the
                                    busy / idle time between the first and the last ggms launch
    <round>_pmc_hbm_traffic.csv     FETCH_SIZE / WRITE_SIZE per kernel (raw KB, separate passes)
    <round>_extract_traffic_<preset>.json   the gather kernel's HBM bytes per launch, corrected as
                                    /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes;
                                    bench.py reads it for roofline.traffic
"""
import collections
import csv
import glob
import json
import os
import sys


def one(pattern):
    """The file of the BENCH process: the default command also runs the `engine` sub-record in a child process, which
    rocprofv3 traces into files of its own (another pid).  Only the bench process launches k_fabric_probe
    (roofline_sampler) and the IdentRows gather; the child is recognised by their absence."""
    # newest first: gpurun MERGES a call's files into the local gpurun_out/, so an earlier collection's files may lie beside
    f = sorted(glob.glob(pattern), key=os.path.getmtime, reverse=True)
    if not f:
        raise SystemExit(f"missing {pattern}")
    for cand in f:
        if "k_fabric_probe" in open(cand, errors="replace").read():
            return cand
    return f[0]


def main():
    src, dst = sys.argv[1], sys.argv[2]
    launches = int(sys.argv[3]) if len(sys.argv) > 3 else 63  # steps of the main region (warm-up 3 + 3 blocks x 20)
    os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)

    # ---- rocprofv3 --stats table, filtered
    rows = list(csv.reader(open(one(f"{src}/trace/*/*_kernel_stats.csv"))))
    with open(f"{dst}_kernel_stats.csv", "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(rows[0])
        for r in rows[1:]:
            if "ggms" in r[0] or "Copy" in r[0] or "copy" in r[0]:
                w.writerow(r)

    # ---- timeline: per-kernel durations + queue gaps over the region that holds the ggms launches
    tr = list(csv.DictReader(open(one(f"{src}/trace/*/*_kernel_trace.csv"))))
    # the main timed region ends where the host-tier sub-record starts: its gather (PlainRows, 16-byte chunks) reads
    # pinned host memory and takes milliseconds -- keep the timeline to what happens before the first such launch
    host_tier = [int(r["Start_Timestamp"]) for r in tr if "k_gather_rows<16, ggms::PlainRows" in r["Kernel_Name"]
                 and int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 2_000_000]
    t_end = min(host_tier) if host_tier else None
    per = collections.OrderedDict()
    queues = collections.defaultdict(list)
    for r in tr:
        if "ggms" not in r["Kernel_Name"]:
            continue
        if t_end is not None and int(r["Start_Timestamp"]) >= t_end:
            continue
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        per.setdefault(r["Kernel_Name"], []).append((e - s) / 1e3)
        queues[r["Queue_Id"]].append((s, e))
    with open(f"{dst}_kernel_timeline.csv", "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(["kernel", "launches", "launches_per_step", "avg_us", "min_us", "max_us", "us_per_step"])
        for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([k, len(v), round(len(v) / launches, 2), round(sum(v) / len(v), 2), round(min(v), 2),
                        round(max(v), 2), round(sum(v) / launches, 2)])
        w.writerow([])
        w.writerow(["queue", "ggms_launches", "span_us", "busy_us", "idle_us", "idle_per_launch_us"])
        for q, iv in sorted(queues.items()):
            iv.sort()
            # drop the warm-up / one-off launches: keep the densest tail (last 80 % of launches)
            iv = iv[len(iv) // 5:]
            span = (iv[-1][1] - iv[0][0]) / 1e3
            busy = sum(e - s for s, e in iv) / 1e3
            w.writerow([q, len(iv), round(span, 1), round(busy, 1), round(span - busy, 1),
                        round((span - busy) / max(1, len(iv) - 1), 2)])

    # ---- PMC passes
    pmc = collections.OrderedDict()
    for sub in ("pmc_fetch", "pmc_write"):
        for r in csv.DictReader(open(one(f"{src}/{sub}/*/*_counter_collection.csv"))):
            if "ggms" not in r["Kernel_Name"]:
                continue
            pmc.setdefault((r["Kernel_Name"], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    with open(f"{dst}_pmc_hbm_traffic.csv", "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(["kernel", "counter", "dispatches", "avg_value_KB", "max_value_KB"])
        for (k, c), v in sorted(pmc.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([k, c, len(v), sum(v) / len(v), max(v)])

    # ---- the gather kernel of the main timed region (IdentRows: full cache in node order): bytes per launch
    bench = json.loads(open(f"{src}/bench_trace.json").read().strip().splitlines()[-1])
    preset = bench["config"]["workload"].split("-shaped")[0]
    rows_per_launch = bench["per_gpu"]["rows_per_step"]
    row_bytes = bench["roofline"]["algorithmic_bytes_per_row"]
    is_main = lambda k: "k_gather_rows" in k and ("IdentRows" in k or "CachedRows" in k)  # noqa: E731
    gk = [k for (k, c) in pmc if is_main(k) and c == "FETCH_SIZE"]
    out = {"kernel": gk[0] if gk else None,
           "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE -- python bench.py [--no-cpu-baseline]  (separate passes)",
           "rows_per_launch": rows_per_launch}
    if gk:
        fetch = pmc[(gk[0], "FETCH_SIZE")]
        write = pmc[(gk[0], "WRITE_SIZE")]
        out["FETCH_SIZE_KB_raw"] = sum(fetch) / len(fetch)
        out["WRITE_SIZE_KB"] = sum(write) / len(write)
        out["fetch_correction"] = ("x2 (MI355X_MICROARCH.md HBM section: gfx950 FETCH_SIZE tallies 128-B requests at "
                                   "64 B for 16-B/lane loads)")
        out["hbm_bytes_per_launch"] = (2 * out["FETCH_SIZE_KB_raw"] + out["WRITE_SIZE_KB"]) * 1024
        out["algorithmic_bytes_per_launch"] = rows_per_launch * row_bytes
        out["traffic_over_algorithmic"] = out["hbm_bytes_per_launch"] / out["algorithmic_bytes_per_launch"]
    # the same kernel's durations in the kernel trace: warm-up + repeats x steps launches inside the pipeline, then
    # 10 launches with nothing beside it (roofline.frac_alone)
    g = [(int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in tr
         if is_main(r["Kernel_Name"])]
    g.sort()
    durs = [d for _, d in g]
    # (before the main region: the trial of sampling pipelines x extract streams, config.pipelines_trial)
    pt = bench["config"].get("pipelines_trial") or {}
    skip = len(pt.get("ms_per_step", {})) * (pt.get("steps_each", 0) * pt.get("blocks_each", 1) + pt.get("warmup_each", 0))
    g, durs = g[skip:], durs[skip:]
    in_pipe = bench["warmup"] + bench["steps"] * bench["repeats"]["blocks"]
    if len(durs) >= in_pipe + 10:
        timed = durs[bench["warmup"]:in_pipe]
        out["rocprof_avg_us_timed_steps"] = sum(timed) / len(timed)
        out["rocprof_avg_us_alone"] = sum(durs[in_pipe:in_pipe + 10]) / 10
        # launches in flight per timed block: sum of durations / (first start -> last end); two extract streams overlap them
        fl = []
        for blk in range(bench["repeats"]["blocks"]):
            lo = bench["warmup"] + blk * bench["steps"]
            seg = g[lo:lo + bench["steps"]]
            span = (seg[-1][0] - seg[0][0]) / 1e3 + seg[-1][1]
            fl.append(sum(d for _, d in seg) / span)
        out["rocprof_launches_in_flight"] = sum(fl) / len(fl)
        out["bench_launches_in_flight_same_run"] = bench["roofline"].get("launches_in_flight")
        out["extract_streams"] = bench["roofline"].get("extract_streams")
        out["bench_avg_launch_us_same_run"] = bench["roofline"]["avg_launch_us"]
        out["bench_avg_launch_us_alone_same_run"] = bench["roofline"]["avg_launch_us_alone"]
    # what the counters describe: bench.py nulls roofline.traffic when the kernel source has changed since
    import hashlib
    src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "xgnn_amd", "csrc", "extract.hip")
    out["extract_hip_sha256"] = hashlib.sha256(open(src, "rb").read()).hexdigest()
    with open(f"{dst}_extract_traffic_{preset}.json", "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))
    print(open(f"{dst}_kernel_timeline.csv").read())

    # ---- the sampler chain: bytes at the L2's memory side per batch, from the same PMC passes
    # batches of the run = launches of the seed insert (one per batch, every sampler launch of the run included);
    # FETCH_SIZE is doubled for the kernels whose reads are wide coalesced streams (the guide's gfx950 note: 128-byte
    # requests tallied at 64 B) and taken as it is for the samplers' 4-byte neighbour / indptr loads (one 64-byte
    # request each, counted exactly -- the probe kernels in the same file: 3.37 M random loads = 209 MB = 64 B each).
    # Atomics show up as 32 bytes of WRITE_SIZE each and no FETCH_SIZE: the table line's read and write-back at the
    # DRAM are not in these counters.
    def per_batch(match, counter, corr):
        ks = [k for (k, c) in pmc if c == counter and match in k]
        return sum(sum(pmc[(k, counter)]) for k in ks) * corr * 1024
    # batches of the run: a batch starts with the first layer's launch that also enters the distinct seeds
    # (k_khop3_fused<.., true, true>) or, on the general path, with the seeds' insert (k_ht_insert) / k_seed_enter
    batches = sum(len(v) for (k, c), v in pmc.items() if c == "FETCH_SIZE" and
                  ("k_ht_insert" in k or "k_seed_enter" in k or ("k_khop3_fused<" in k and ", true, true>" in k)))
    if batches:
        parts = {}
        for name, match, corr in (("fused samplers", "k_khop3_fused", 1.0), ("owner scans", "k_owner_scan", 2.0),
                                  ("look-ups", "k_map_rest_all", 2.0), ("seed insert", "k_ht_insert", 1.0),
                                  ("seed enter", "k_seed_enter", 1.0)):
            parts[name] = {"fetch_bytes": per_batch(match, "FETCH_SIZE", corr) / batches,
                           "write_bytes": per_batch(match, "WRITE_SIZE", 1.0) / batches, "fetch_correction": corr}
        total = sum(p["fetch_bytes"] + p["write_bytes"] for p in parts.values())
        E, S = bench["per_gpu"]["edges_per_step"], (bench.get("roofline_sampler") or {}).get("seeds_per_step")
        st = {"batches": batches, "per_kernel_group": parts, "hbm_bytes_per_batch": total,
              "edges_per_batch": E, "seeds_per_batch": S,
              "algorithmic_bytes_per_batch": (12 * S + 28 * E) if S else None,
              "traffic_over_algorithmic": total / (12 * S + 28 * E) if S else None,
              "note": "64-byte lines for the 4-byte neighbour and indptr reads and 32-byte atomic payloads against SURVEY "
                      "8d's 4 + 16 bytes per edge; the dedup table's line read and write-back per atomic are not counted here"}
        srcs = ["sample_khop.hip", "hashtable.hip", "sample_batch.hip", "ggms_device.h", "tile_scan.h"]
        root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "xgnn_amd", "csrc")
        st["sources_sha256"] = hashlib.sha256(b"".join(open(os.path.join(root, f), "rb").read() for f in srcs)).hexdigest()
        st["sources"] = srcs
        with open(f"{dst}_sampler_traffic_{preset}.json", "w") as f:
            json.dump(st, f, indent=1)
        print(json.dumps(st, indent=1))


if __name__ == "__main__":
    main()
