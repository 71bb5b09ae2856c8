"""CPU suite: bench.py's LAST stdout line must stay small enough for the driver to parse (round 5's one ~28 KB line could not
be parsed and the round lost its record).  bench_legs/line.py builds it from the full object; the full object travels on the line before
it (`DETAIL {...}`)."""
import io
import json
import os

import pytest

from bench_legs import line as BL

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
            "config", "roofline", "cpu_baseline")


def _canned():
    """the committed full objects of earlier rounds (the round-5 one is the line that broke the record)"""
    out = []
    for name in sorted(os.listdir(os.path.join(ROOT, "profiles"))):
        if name.endswith("_bench_n1.json"):
            txt = open(os.path.join(ROOT, "profiles", name)).read()
            d = BL.detail_of(txt)
            if d is not None:
                out.append((name, d))
    return out


def _strings(x):
    if isinstance(x, dict):
        for v in x.values():
            yield from _strings(v)
    elif isinstance(x, list):
        for v in x:
            yield from _strings(v)
    elif isinstance(x, str):
        yield x


@pytest.mark.parametrize("name,full", _canned(), ids=[n for n, _ in _canned()])
def test_compact_line_of_committed_bench_objects(name, full):
    c = BL.compact_line(full)
    s = json.dumps(c, ensure_ascii=True, allow_nan=False)
    assert len(s) < BL.MAX_LINE, (name, len(s))
    assert s.isascii() and "\n" not in s
    for k in REQUIRED:
        assert k in c, (name, k)
    assert c["value"] == full["value"] and c["ms_per_step"] == full["ms_per_step"] and c["n_gpus"] == full["n_gpus"]
    assert c["config"]["workload"] and "model" not in c["config"]
    rf = c["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, (name, k)
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-4
    cb = c["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, (name, k)
    assert all(len(t) <= BL.MAX_STR for t in _strings(c))


def test_round5_line_shrinks_from_28kb_and_keeps_one_number_per_leg():
    full = dict(_canned())["r5_bench_n1.json"]
    assert len(json.dumps(full)) > 25000                       # the line the driver could not parse
    c = BL.compact_line(full)
    legs = c["legs"]
    assert legs["compact_only"]["value"] == pytest.approx(full["compact_only"]["value"], rel=1e-5)
    assert legs["header_range_1024"]["frac"] == pytest.approx(full["header_range_1024"]["roofline"]["frac"], rel=1e-5)
    assert legs["stress_v100"]["frac"] == pytest.approx(full["stress"]["v100"]["roofline"]["frac"], rel=1e-5)
    assert legs["stress_v512"]["headers_per_s"] == pytest.approx(full["stress"]["v512"]["headers_per_s"], rel=1e-5)
    assert legs["latency"]["coalesced_k16"]["p99_ms"] > 0 and legs["hint_burst_32"]["median_ms"] > 0
    assert legs["fused_commitment"]["frac"] == pytest.approx(full["fused_commitment"]["roofline"]["frac"], rel=1e-5)
    assert c["long_run"]["steps"] == 200 and c["long_run"]["seconds"] > 0.1
    assert c["roofline"]["isolated_frac"] == pytest.approx(full["roofline"]["isolated"]["frac"], rel=1e-5)
    assert c["roofline"]["frac_survey_8d"] == pytest.approx(full["roofline"]["byte_accounting"]["frac_survey_8d"], rel=1e-5)


def test_nan_long_strings_and_oversized_legs_cannot_break_the_line():
    full = dict(_canned())["r5_bench_n1.json"]
    full = json.loads(json.dumps(full))
    full["roofline"]["traffic"] = float("nan")
    full["cpu_baseline"]["sample"] = "x" * 5000 + " § non-ascii \t tab"
    full["config"]["collective"] = "y" * 1000
    full["config"]["multi_gpu"] = {"per_rank_ms_per_step": [5.5] * 8, "allgather_us_per_chunk": {"avg": [30.0] * 8, "max": [99.0] * 8, "min": [1.0] * 8},
                                   "note": "z" * 3000}
    # a future leg that explodes: the line must shed legs, not the contract
    full["latency"]["concurrent"]["coalesced_shared_context"] = [dict(r, threads=k) for k, r in
                                                                 zip((1, 16, 64), full["latency"]["concurrent"]["coalesced_shared_context"][:3])]
    full["keyset_churn"]["by_rotate_permille"] = [{"rotate_permille": i, "ms_per_step": 0.123456789} for i in range(400)]
    buf = io.StringIO()
    BL.emit(full, file=buf)
    lines = buf.getvalue().splitlines()
    assert len(lines) == 2 and lines[0].startswith(BL.DETAIL_PREFIX) and lines[1].startswith("{")
    assert len(lines[1]) < BL.MAX_LINE and lines[1].isascii() and "NaN" not in lines[1] and "NaN" not in lines[0]
    c = json.loads(lines[1])
    assert c["roofline"]["traffic"] is None and len(c["cpu_baseline"]["sample"]) <= BL.MAX_STR
    assert "keyset_churn_ms" in c["legs_dropped_for_size"] and "compact_only" in c["legs"]
    for k in REQUIRED:
        assert k in c
    assert c["config"]["multi_gpu"]["per_rank_ms_per_step"] == [5.5] * 8
    # the DETAIL line round-trips to the full object (NaN -> null)
    back = BL.detail_of(buf.getvalue())
    assert back["value"] == full["value"] and back["roofline"]["traffic"] is None
    assert back["stress"]["v512"]["range_verdict"] == full["stress"]["v512"]["range_verdict"]


def test_mode_s_primary_object():
    """--mode S prints the stress object as the primary one: same two lines"""
    v = dict(_canned())["r5_bench_n1.json"]["stress"]["v512"]
    full = {"metric": "headers/sec, mode S (a commit on every header)", "value": v["headers_per_s"], "unit": "headers/s", "n_gpus": 1, "steps": 5, "warmup": 1,
            "ms_per_step": v["ms"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": v["workload"], "nccl_ranks": 1, "dist_backend": None, "parallelism": "1 x 2048 commits"},
            "roofline": v["roofline"], "cpu_baseline": v["cpu_baseline"], "stress": v}
    c = BL.compact_line(full)
    assert len(json.dumps(c)) < BL.MAX_LINE
    assert c["roofline"]["bound"] == "valu" and c["roofline"]["valu_issue_frac"] > 0 and c["cpu_baseline"]["cores"] == 32
    assert c["legs"]["stress"]["all_ok"] is True and c["legs"]["stress"]["keytable_MB"] > 0


def test_bench_py_prints_through_emit_only():
    """no bare print(json.dumps(out)) of a full object may come back into bench.py"""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "emit(out)" in src and "print(json.dumps(out))" not in src
    assert "detail_of(out.stdout)" in src              # the sub-process legs read the DETAIL line


def test_design_numbers_are_the_profile_s():
    """every number of DESIGN.md §5 comes from profiles/r6_bench_n1.json — the table is generated
    (tools/design_numbers.py) and the committed block must be what the generator produces from the committed profile."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("design_numbers", os.path.join(ROOT, "tools", "design_numbers.py"))
    dn = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(dn)
    full = json.load(open(dn.SRC))
    text = open(os.path.join(ROOT, "DESIGN.md")).read()
    i, j = text.index(dn.BEGIN), text.index(dn.END)
    assert text[i + len(dn.BEGIN):j].strip() == dn.block(full).strip(), "run `python tools/design_numbers.py` after replacing profiles/r6_bench_n1.json"
    assert len(text.splitlines()) <= 300, "DESIGN.md is the current state only (<= 300 lines); history goes to HISTORY.md"
    # no live citation of an older round's profile in the current-state document
    import re
    assert not re.findall(r"profiles/r[1-5]_", text), re.findall(r"profiles/r[1-5]_\w+", text)
