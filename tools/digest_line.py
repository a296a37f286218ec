"""bench.py's JSON line -> the short text records kept under profiles/ (configs, staged host tier, box, xgmi).

    python tools/digest_line.py <bench.json> [configs|staged|box|xgmi|all]
"""
import json
import sys


def main():
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    what = sys.argv[2] if len(sys.argv) > 2 else "all"
    if what in ("box", "all") and d.get("box"):
        b = d["box"]
        print(f"box: d2d copy {b['d2d_copy_GBps']:.0f} GB/s (read + written; {b['d2d_over_guide']:.3f} of the guide's 6290), pinned H2D "
              f"{b['pinned_h2d_GBps']:.1f} GB/s, atomics {b.get('atomics_per_s', 0) / 1e9:.1f} G/s, random 4-B loads "
              f"{b.get('loads_per_s', 0) / 1e9:.1f} G/s | headline {d['value']:.4g} edges/s, {d['ms_per_step']:.4f} ms/step, "
              f"value_over_box {d['value_over_box']:.4g}")
        pt = d.get("config", {}).get("pipelines_trial")
        if pt:
            print(f"streams: {pt['chosen']['pipelines']} sampling pipeline(s), {pt['chosen']['extract_streams']} extract stream(s) "
                  "(trial before the timed region, ms/step: " + ", ".join(f"{k}: {v:.4f}" for k, v in pt["ms_per_step"].items()) + ")")
        r = d["roofline"]
        if r.get("launches_in_flight", 1.0) > 1.0:
            print(f"gather launches overlap on {r['extract_streams']} extract streams: {r['launches_in_flight']:.2f} in flight on average, a "
                  f"launch's own duration {r['avg_launch_us']:.0f} us (one launch alone gets {r['frac_of_one_launch']:.3f} of 8 TB/s)")
        print(f"gather: {r['frac']:.3f} of 8 TB/s in the pipeline, {r['frac_alone']:.3f} alone; hbm_read_frac {r['hbm_read_frac']:.3f} "
              f"(target {r['hbm_read_frac_target']}, ceiling for a copy on this box {r['hbm_read_frac_ceiling_for_a_copy']:.3f})")
    if what in ("configs", "all") and d.get("configs"):
        for k, c in d["configs"].items():
            if "ms_per_step" not in c:
                print(f"configs.{k}: {c}")
                continue
            print(f"configs.{k}: {c['ms_per_step']:.4f} ms/step, {c['edges_per_s']:.4g} edges/s, feature {c['feature_extract_GBps']:.0f} GB/s, "
                  f"gather {c['gather_frac']:.3f} in the pipeline / {c['gather_frac_alone']:.3f} alone, {c['edges_per_step']:.0f} edges and "
                  f"{c['rows_per_step']:.0f} rows per step"
                  + (f"; {c['pipelines_trial']['chosen']['pipelines']} sampling pipeline(s), "
                     f"{c['pipelines_trial']['chosen']['extract_streams']} extract stream(s) (trial: "
                     + ", ".join(f"{v:.4f}" for v in c["pipelines_trial"]["ms_per_step"].values()) + " ms/step)"
                     if c.get("pipelines_trial") else "")
                  + (f"; through samgraph.torch (engine): {c['engine']['ms_per_step']:.4f} ms/step, {c['engine']['edges_per_s']:.4g} edges/s"
                     if isinstance(c.get("engine"), dict) and "ms_per_step" in c["engine"] else ""))
    st = (d.get("host_tier") or {}).get("staged") or d.get("host_tier_staged")
    if what in ("staged", "all") and st:
        h = d.get("host_tier") or {}
        if "feature_extract_GBps" in h:
            print(f"host tier, zero-copy (gpu_extract on): {h['feature_extract_GBps']:.1f} GB/s = {h['frac_of_pinned_copy']:.3f} of the box's "
                  f"pinned H2D copy ({h['pinned_h2d_copy_GBps']:.1f} GB/s), {h['ms_per_step']:.2f} ms/step")
        ref = st["reference_published"]
        print(f"reference, published ({ref['source']}): CPU gather {ref['cpu_gather_GBps'][0]}-{ref['cpu_gather_GBps'][1]} GB/s, "
              f"H2D {ref['h2d_GBps']} GB/s, combine {ref['combine_miss_GBps']} GB/s")
        for k in ("cache_0", "cache_0.64"):
            c = st.get(k)
            if not c or "serial" not in c:
                print(f"host-staged {k}: {c}")
                continue
            s, o = c["serial"], c["overlapped"]
            fmt = lambda v: "-" if v is None else f"{v:.1f}"  # noqa: E731
            print(f"host-staged {k} ({c['host_threads']} host threads, {s['miss_MB_per_step']:.0f} MB of miss rows per step):")
            print(f"  serial (the reference's sequence, each phase behind its own wait): CPU gather {fmt(s['cpu_gather_GBps'])} GB/s, "
                  f"H2D {fmt(s['h2d_GBps'])} GB/s, combine-miss {fmt(s['combine_miss_GBps'])} GB/s, combine-cache "
                  f"{fmt(s['combine_cache_GBps'])} GB/s -> {fmt(s['effective_GBps'])} GB/s effective, {s['ms_per_step']:.2f} ms/step")
            print(f"  chunked pipeline: {o['effective_GBps']:.1f} GB/s effective over the wall clock ({o['ms_per_step']:.2f} ms/step, sampling "
                  f"+ split + hit combine included) = {o['over_min_of_cpu_gather_and_h2d']:.2f} x min(CPU gather, H2D) of the serial "
                  f"phases; host gather busy rate {fmt(o['cpu_gather_busy_GBps'])} GB/s; {o['edges_per_s']:.3g} sampled edges/s")
    if what in ("xgmi", "all") and d.get("xgmi"):
        x = d["xgmi"]
        print("peer_access:", d.get("peer_access", {}).get("can_access"), "refused", d.get("peer_access", {}).get("refused"))
        if "error" in x:
            print("xgmi:", x)
        else:
            print(f"xgmi probe ({x['seconds']} s, {x['probe_bytes'] >> 20} MiB per rank, {x['row_bytes']}-B rows) [reader][owner]:")
            for name in ("per_pair_copy_GBps", "per_pair_stream_kernel_GBps", "per_pair_gather_GBps"):
                print(" ", name)
                for row in x[name]:
                    print("   ", " ".join(f"{v:8.1f}" for v in row))
            print("  inbound, all peers at once:", " ".join(f"{v:.1f}" for v in x["inbound_all_peers_gather_GBps"]))
        for k, s in (d.get("stores") or {}).items():
            if isinstance(s, dict) and "ms_per_step" in s:
                print(f"stores.{k}: {s['ms_per_step']:.4f} ms/step, remote rows {s['remote_row_fraction']:.3f}, xGMI bytes per GPU and step "
                      f"{s.get('xgmi_bytes_per_step_per_gpu', 0) / 1e6:.1f} MB, predicted remote {s.get('predicted_remote_ms_per_step')} ms"
                      + (f", replicated {s['replicated_fraction']:.3f} ({s.get('hbm_spent_gb', 0):.1f} GB)" if "replicated_fraction" in s else ""))


if __name__ == "__main__":
    main()
