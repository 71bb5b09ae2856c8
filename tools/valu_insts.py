#!/usr/bin/env python3
"""profiles/valu_insts.json from the SQ counter summaries (tools/pmc_summary.py output of `rocprofv3 --pmc ... SQ_INSTS_VALU`
passes): wave-level VALU instructions per UNIT OF WORK of the ALU-bound kernels, which bench.py turns into
valu_issue_frac = instructions / (time x measured v_add_u32 wave-issue rate).
usage: valu_insts.py out.json  name=csv:kernel_substr:units_per_launch:unit_label ...   (per_launch_max of SQ_INSTS_VALU is used)"""
import csv, json, sys
out = {"note": "SQ_INSTS_VALU counts wave-level instructions; wave_valu_insts_per_unit = per_launch_max / (lane-level) units of that launch; "
               "x 64 = VALU instructions one wave executes per unit in each of its lanes (valu_insts_per_unit_per_wave)", "kernels": {}}
for spec in sys.argv[2:]:
    name, rest = spec.split("=", 1)
    path, substr, units, label = rest.split(":", 3)
    best = None
    for r in csv.DictReader(open(path)):
        if r["counter"] == "SQ_INSTS_VALU" and substr in r["kernel"]:
            v = int(r["per_launch_max"])
            best = (v, r["kernel"]) if best is None or v > best[0] else best
    if best:
        out["kernels"][best[1]] = {"wave_valu_insts_per_unit": best[0] / float(units), "unit": label, "per_launch": best[0],
                                   "units_per_launch": float(units), "source": path.split("/")[-1], "alias": name,
                                   "valu_insts_per_unit_per_wave": 64 * best[0] / float(units)}
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
