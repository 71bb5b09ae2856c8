"""The two stdout lines of bench.py.

VERDICT r5 #1: round 5's ONE ~28 KB JSON line could no longer be parsed by the driver, and the round lost its record.  Now:

    DETAIL {...every leg, every note...}          <- the line before last: the full object (profiles/rN_bench_n1.json is this)
    {...compact...}                               <- the LAST line: the contract's keys + roofline + cpu_baseline + one number per leg

The compact line is ASCII, has no NaN/Infinity, every string is <= 120 characters and the whole line is < 6 KB
(tests/test_bench_line.py holds it to that on the committed round-5 object and on synthetic worst cases).
"""
import json
import math

MAX_LINE = 6144
MAX_STR = 120
DETAIL_PREFIX = "DETAIL "

CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                 "dtype", "data")
CONFIG_KEYS = ("workload", "timed_entry", "ranges_per_gpu", "headers_per_step", "pipelined_chunks", "parallelism", "nccl_ranks", "dist_backend",
               "collective", "witness_checked_ranges", "witness_bytes_per_step_per_gpu", "ed25519_path")
ROOFLINE_KEYS = ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "avg_launch_ms", "algorithmic_bytes_per_launch",
                 "launches_timed", "frac_of_measured_store_ceiling")
CPU_KEYS = ("value", "unit", "cores", "kind", "sample", "cpu_model", "verifies_per_s")


def _num(x, digits=6):
    """floats to `digits` significant figures; NaN / inf -> None (JSON has neither)"""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    if isinstance(x, float):
        if math.isnan(x) or math.isinf(x):
            return None
        return float(f"{x:.{digits}g}")
    return x


def _s(x):
    if isinstance(x, str):
        x = x.encode("ascii", "replace").decode("ascii").replace("\t", " ")
        x = " ".join(x.split())
        return x if len(x) <= MAX_STR else x[:MAX_STR - 3] + "..."
    return x


def _clean(x):
    if isinstance(x, dict):
        return {str(k): _clean(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_clean(v) for v in x]
    return _s(_num(x))


def _get(d, *path, default=None):
    for k in path:
        if isinstance(d, dict):
            d = d.get(k)
        elif isinstance(d, (list, tuple)) and isinstance(k, int) and -len(d) <= k < len(d):
            d = d[k]
        else:
            return default
        if d is None:
            return default
    return d


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


def _row(rows, key, val):
    for r in rows or []:
        if isinstance(r, dict) and r.get(key) == val:
            return r
    return None


def _roofline(rf):
    if not isinstance(rf, dict):
        return None
    out = _pick(rf, ROOFLINE_KEYS)
    if "byte_accounting" in rf:
        out["frac_survey_8d"] = _get(rf, "byte_accounting", "frac_survey_8d")
    if "isolated" in rf:
        out["isolated_frac"] = _get(rf, "isolated", "frac")
        out["isolated_launch_ms"] = _get(rf, "isolated", "avg_launch_ms")
    if "valu_issue" in rf:
        out["valu_issue_frac"] = _get(rf, "valu_issue", "valu_issue_frac")
    return out


def _cpu(cb):
    return _pick(cb, CPU_KEYS) if isinstance(cb, dict) else None


def _stress(sv):
    """one mode-S object -> its few numbers"""
    if not isinstance(sv, dict) or "error" in sv:
        return sv if sv is None else {"error": _get(sv, "error")}
    return {"headers_per_s": sv.get("headers_per_s"), "ms": sv.get("ms"), "one_in_flight_ms": _get(sv, "one_step_in_flight", "ms"),
            "frac": _get(sv, "roofline", "frac"), "valu_issue_frac": _get(sv, "roofline", "valu_issue", "valu_issue_frac"),
            "verify_ms": _get(sv, "stage_ms", "ed25519_verify_keyed"), "tally_ms": _get(sv, "stage_ms", "tally_validator_hash"),
            "sha512_ms": _get(sv, "stage_ms", "sha512_challenge"),
            "tally_frac": _get(sv, "roofline", "commit_tally", "frac_of_measured_peak"),
            "witness_headers_per_s": _get(sv, "witness", "headers_per_s"), "witness_frac": _get(sv, "witness", "roofline", "frac"),
            "witness_traffic": _get(sv, "witness", "roofline", "traffic"),
            "keytable_MB": sv.get("keytable_MB"), "all_ok": _get(sv, "range_verdict", "all_ok"),
            "cpu_headers_per_s": _get(sv, "cpu_baseline", "value")}


def _legs(full):
    """<= a handful of numbers per secondary leg (the DETAIL line has the rest)"""
    legs = {}
    co = full.get("compact_only")
    if isinstance(co, dict):
        legs["compact_only"] = {"value": co.get("value"), "ms_per_step": co.get("ms_per_step"),
                                "frac_sha256_whole_step": co.get("frac_of_measured_alu_peak_whole_step"),
                                "frac_sha256_prove_subchain": co.get("frac_of_measured_alu_peak_prove_subchain"),
                                "prove_subchain_ms": co.get("prove_subchain_ms")} if "error" not in co else {"error": co["error"]}
    h = full.get("header_range_1024")
    if isinstance(h, dict):
        legs["header_range_1024"] = {"value": h.get("value"), "ms_per_step": h.get("ms_per_step"), "frac": _get(h, "roofline", "frac"),
                                     "traffic": _get(h, "roofline", "traffic"), "avg_launch_ms": _get(h, "roofline", "avg_launch_ms"),
                                     "algorithmic_bytes_per_launch": _get(h, "roofline", "algorithmic_bytes_per_launch")} if "error" not in h else {"error": h["error"]}
    st = full.get("stress")
    if isinstance(st, dict):
        if "v100" in st or "v512" in st:
            legs["stress_v100"], legs["stress_v512"] = _stress(st.get("v100")), _stress(st.get("v512"))
        else:
            legs["stress"] = _stress(st)                  # --mode S as the primary object
    lat = full.get("latency")
    if isinstance(lat, dict):
        L = {"single_range_ms": _get(lat, "output_only_ms", "median"), "next_header_ms": _get(lat, "next_header_ms", "median")}
        conc = lat.get("concurrent")
        if isinstance(conc, dict) and "error" in conc:
            L["concurrent_error"] = str(conc["error"])[-110:]       # a leg that failed must be visible in the record
        if isinstance(conc, dict):
            for name, rows_key in (("coalesced", "coalesced_shared_context"), ("page_locked", "coalesced_page_locked"),
                                   ("registered", "coalesced_registered_once"), ("packed", "coalesced_packed_headers")):
                for k in (1, 16, 64):
                    r = _row(conc.get(rows_key), "threads", k)
                    if r:
                        L[f"{name}_k{k}"] = {"headers_per_s": r.get("headers_per_s"), "p99_ms": r.get("p99_ms")}
        legs["latency"] = L
    hc = full.get("hint_concurrent")
    if isinstance(hc, dict) and "error" in hc:
        legs["hint_burst_32"] = {"error": str(hc["error"])[-110:]}
    elif isinstance(hc, dict):
        legs["hint_burst_32"] = {"median_ms": _get(hc, "coalesced", "hint_only", "median_ms"),
                                 "map_job_one_call_median_ms": _get(hc, "coalesced", "map_job_one_call", "median_ms"),
                                 "serial_median_ms": _get(hc, "serial", "hint_only", "median_ms")}
    fc = full.get("fused_commitment")
    if isinstance(fc, dict):
        legs["fused_commitment"] = {"frac": _get(fc, "roofline", "frac"), "G_perm_per_s": _get(fc, "roofline", "achieved"),
                                    "valu_issue_frac": _get(fc, "roofline", "valu_issue", "valu_issue_frac"),
                                    "caps_mode_headers_per_s": _get(fc, "pipeline_caps_mode", "headers_per_s")}
    up = full.get("with_input_upload")
    if isinstance(up, dict):
        legs["with_input_upload"] = {"value": up.get("value"), "h2d_GBps": up.get("h2d_GBps")}
    ab = full.get("units_ab")
    if isinstance(ab, dict):
        legs["units_ab"] = {"with_units_ms": _get(ab, "with_units", "expand_map_avg_launch_ms"),
                            "without_units_ms": _get(ab, "without_units", "expand_map_avg_launch_ms"),
                            "with_units_value": _get(ab, "with_units", "value")}
    kc = full.get("keyset_churn")
    if isinstance(kc, dict):
        rows = kc.get("by_rotate_permille") or []
        legs["keyset_churn_ms"] = {str(r.get("rotate_permille")): r.get("ms_per_step") for r in rows if isinstance(r, dict)}
    rs = full.get("range_sweep")
    if isinstance(rs, dict):
        r1 = _row(_get(rs, "witness", "by_ranges"), "ranges", 1)
        if r1:
            legs["range_sweep_R1_witness_headers_per_s"] = r1.get("headers_per_s")
    return {k: v for k, v in legs.items() if v is not None}


def compact_line(full):
    """The compact dict of a full bench object (mode F headline, --mode S, N >= 1)."""
    out = {k: full.get(k) for k in CONTRACT_KEYS}
    cfg = full.get("config") or {}
    c = _pick(cfg, CONFIG_KEYS)
    mg = cfg.get("multi_gpu")
    if isinstance(mg, dict):
        c["multi_gpu"] = {"per_rank_ms_per_step": mg.get("per_rank_ms_per_step"),
                          "allgather_us_per_chunk_avg": _get(mg, "allgather_us_per_chunk", "avg"),
                          "allgather_us_per_chunk_max": _get(mg, "allgather_us_per_chunk", "max")}
    sc = cfg.get("sharded_vs_unsharded_self_check_per_rank")
    if sc is not None:
        c["sharded_vs_unsharded_self_check_per_rank"] = sc
    out["config"] = c
    out["roofline"] = _roofline(full.get("roofline"))
    out["cpu_baseline"] = _cpu(full.get("cpu_baseline"))
    if full.get("long_run") is not None:
        out["long_run"] = _pick(full["long_run"], ("steps", "seconds", "ms_per_step", "value"))
    ws = full.get("roofline_whole_step")
    if isinstance(ws, dict):
        out["roofline_whole_step"] = _pick(ws, ("achieved", "frac", "stored_bytes_per_step", "frac_of_measured_store_ceiling"))
    cal = full.get("calibration")
    if isinstance(cal, dict):
        out["calibration"] = _pick(cal, ("sha256_compress_per_s", "sha512_compress_per_s", "fe25519_mul_per_s", "goldilocks_mul_per_s", "hbm_store_bytes_per_s"))
    k0 = _get(full, "kernels", 0)
    if isinstance(k0, dict):
        out["sha_kernel"] = {"kernel": k0.get("kernel"), "avg_launch_ms": k0.get("avg_launch_ms"), "compact_GBps": k0.get("achieved_GBps"),
                             "frac_of_hbm_peak": k0.get("frac_of_hbm_peak"), "frac_of_measured_alu_peak": k0.get("frac_of_measured_alu_peak")}
    legs = _legs(full)
    if legs:
        out["legs"] = legs
    out["detail"] = "full object: previous stdout line, prefix 'DETAIL '"
    out = _clean(out)
    for k in ("value", "ms_per_step"):                 # the driver cross-checks these against its own clock: unrounded
        if isinstance(full.get(k), float) and math.isfinite(full[k]):
            out[k] = full[k]
    # belt and braces: if some future leg blows the budget, drop legs (largest first) rather than lose the record
    while len(json.dumps(out, allow_nan=False)) >= MAX_LINE and out.get("legs"):
        biggest = max(out["legs"], key=lambda k: len(json.dumps(out["legs"][k])))
        del out["legs"][biggest]
        out["legs_dropped_for_size"] = out.get("legs_dropped_for_size", []) + [biggest]
    return out


def _no_nan(x):
    if isinstance(x, dict):
        return {k: _no_nan(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_no_nan(v) for v in x]
    if isinstance(x, float) and (math.isnan(x) or math.isinf(x)):
        return None
    return x


def emit(full, file=None):
    """print the DETAIL line, then the compact line (LAST on stdout)"""
    import sys
    f = file or sys.stdout
    print(DETAIL_PREFIX + json.dumps(_no_nan(full), ensure_ascii=True, allow_nan=False), file=f, flush=True)
    line = json.dumps(compact_line(full), ensure_ascii=True, allow_nan=False)
    assert len(line) < MAX_LINE, len(line)
    print(line, file=f, flush=True)


def detail_of(stdout):
    """The full object of a bench.py run from its captured stdout (the DETAIL line; for older outputs the last JSON line)."""
    lines = stdout.splitlines()
    for ln in reversed(lines):
        if ln.startswith(DETAIL_PREFIX):
            return json.loads(ln[len(DETAIL_PREFIX):])
    js = [ln for ln in lines if ln.startswith("{")]
    return json.loads(js[-1]) if js else None
