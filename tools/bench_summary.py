"""tools: one-screen summary of a bench.py JSON line (all legs).  usage: bench_summary.py file.json"""
import json, sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench_legs.line import detail_of
d = detail_of(open(sys.argv[1]).read())
r = d["roofline"]
print("headline %.1f M headers/s  %.3f ms/step  expand avg %.3f ms  frac %.3f (store-ceiling %.3f, isolated %s)  traffic %s" % (
    d["value"] / 1e6, d["ms_per_step"], r.get("avg_launch_ms", 0), r["frac"], r.get("frac_of_measured_store_ceiling", 0), r.get("isolated", {}).get("frac") if isinstance(r.get("isolated"), dict) else r.get("isolated_frac"), r.get("traffic")))
c = d["calibration"]
print("ceilings: sha256 %.2f G/s  store %.2f TB/s  fe_mul %.0f G/s  gl_mul %.2f T/s" % (c["sha256_compress_per_s"] / 1e9, c["hbm_store_bytes_per_s"] / 1e12, c["fe25519_mul_per_s"] / 1e9, c["goldilocks_mul_per_s"] / 1e12))
if "autotune" in d.get("config", {}): print("autotune", d["config"]["autotune"])
if "compact_only" in d:
    c = d["compact_only"]; print("compact %.1f M  %.3f ms  frac %.3f  prove_subchain %.3f ms" % (c["value"] / 1e6, c["ms_per_step"], c["frac_of_measured_alu_peak_whole_step"], c["prove_subchain_ms"]))
if "header_range_1024" in d: print("header_range_1024 %.1f M" % (d["header_range_1024"]["value"] / 1e6))
if "with_input_upload" in d: print("with_input_upload", {k: (round(v / 1e6, 1) if isinstance(v, float) and v > 1e5 else v) for k, v in d["with_input_upload"].items() if k in ("value", "headers_per_s", "ms_per_step")})
if "latency" in d:
    l = d["latency"]; print("latency", {k: round(v, 4) for k, v in l["output_only_ms"].items()}, "witness dl %.2f ms" % l["with_witness_download_ms"]["median"])
    if "concurrent" in l: print("concurrent", [(x["threads"], round(x["headers_per_s"] / 1e6, 1), round(x["p50_ms"], 2)) for x in l["concurrent"].get("coalesced_shared_context", l["concurrent"].get("by_threads", []))])
if "range_sweep" in d:
    for k in ("compact", "witness"):
        print("sweep", k, [(x["ranges"], round(x["headers_per_s"] / 1e6, 1), round(x["ms_per_step"], 3)) for x in d["range_sweep"][k]["by_ranges"]])
if "fused_commitment" in d:
    f = d["fused_commitment"]; print("poseidon fused %.2f M headers/s (%.2f ms)  frac %.3f  issue %.3f  caps-mode %.2f M  cpu %.0f headers/s" % (
        f["headers_per_s_fused"] / 1e6, f["fused_ms"], f["roofline"]["frac"], (f["roofline"].get("valu_issue") or {}).get("valu_issue_frac", 0), f["pipeline_caps_mode"]["headers_per_s"] / 1e6, f["cpu_baseline"]["value"]))
for v in ("v100", "v512"):
    if "stress" in d and v in d["stress"]:
        s = d["stress"][v]
        print(v, "%.2f M headers/s %.3f ms (%d in flight; one: %.2f M %.3f ms)  verify %.3f ms frac %.3f issue %s  stages %s" % (
            s["headers_per_s"] / 1e6, s["ms"], s["steps_in_flight"], s["one_step_in_flight"]["headers_per_s"] / 1e6, s["one_step_in_flight"]["ms"], s["roofline"]["avg_launch_ms"], s["roofline"]["frac"],
            round((s["roofline"].get("valu_issue") or {}).get("valu_issue_frac", 0), 3), {k: round(x, 3) for k, x in s["stage_ms"].items()}))
        w = s["witness"]; print("   witness %.2f M headers/s %.2f ms (one: %.2f M %.2f ms)  expansion %.3f ms frac %.3f (store-ceiling %.3f)  cpu %.0f headers/s" % (
            w["headers_per_s"] / 1e6, w["ms"], w["one_step_in_flight"]["headers_per_s"] / 1e6, w["one_step_in_flight"]["ms"], w["roofline"]["avg_launch_ms"], w["roofline"]["frac"], w["roofline"]["frac_of_measured_store_ceiling"], s["cpu_baseline"]["value"]))
print("cpu_baseline %.2f M headers/s on %s threads" % (d["cpu_baseline"]["value"] / 1e6, d["cpu_baseline"]["cores"]))
