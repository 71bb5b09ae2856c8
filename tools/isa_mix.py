#!/usr/bin/env python3
"""Static instruction mix of one kernel in a gfx950 assembly listing (hipcc --cuda-device-only -S).

usage: isa_mix.py file.s kernel_substring [label:trips ...]
Splits the kernel at its .LBB labels, prints the VALU / SALU / s_nop / memory mix per block and — with `label:trips`
pairs naming how often each block runs (e.g. LBB0_2:4 for a loop body of four iterations; blocks not named run once) —
the weighted total per kernel invocation.  v_mad_u64_u32 / v_mul_* count as `mul` (half-rate on gfx950).
"""
import re, sys, collections

def classify(op):
    if op.startswith("s_nop"): return "s_nop"
    if op.startswith("v_mad_u64") or op.startswith("v_mul_lo") or op.startswith("v_mul_hi") or op.startswith("v_mad_u32"): return "v_mul"
    if op.startswith("v_mov") or op.startswith("v_accvgpr"): return "v_mov"
    if op.startswith("v_cndmask"): return "v_cndmask"
    if op.startswith("v_cmp"): return "v_cmp"
    if op.startswith("v_"): return "v_other"
    if op.startswith("s_waitcnt"): return "s_wait"
    if op.startswith("s_"): return "salu"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_", "ds_")): return "mem"
    return "other"

def main():
    path, sub = sys.argv[1], sys.argv[2]
    trips = dict((a.split(":")[0], int(a.split(":")[1])) for a in sys.argv[3:])
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*%s\S*:" % re.escape(sub), l))
    blocks, cur = collections.OrderedDict(), "entry"
    blocks[cur] = collections.Counter()
    ops = collections.OrderedDict(); ops[cur] = collections.Counter()
    for l in lines[start + 1:]:
        m = re.match(r"^\.(LBB\d+_\d+):", l)
        if m:
            cur = m.group(1); blocks[cur] = collections.Counter(); ops[cur] = collections.Counter(); continue
        t = l.strip()
        if not t or t.startswith((";", ".")): continue
        op = t.split()[0]
        if not re.match(r"^[a-z]", op): continue
        blocks[cur][classify(op)] += 1
        ops[cur][op] += 1
        if op == "s_endpgm": break
    keys = ["v_mul", "v_other", "v_mov", "v_cndmask", "v_cmp", "s_nop", "salu", "s_wait", "mem"]
    print("%-10s %6s " % ("block", "trips") + " ".join("%9s" % k for k in keys) + "     VALU")
    tot = collections.Counter(); totops = collections.Counter()
    for b, c in blocks.items():
        n = trips.get(b, 1)
        valu = c["v_mul"] + c["v_other"] + c["v_mov"] + c["v_cndmask"] + c["v_cmp"]
        print("%-10s %6d " % (b, n) + " ".join("%9d" % c[k] for k in keys) + " %8d" % valu)
        for k in c: tot[k] += n * c[k]
        for k in ops[b]: totops[k] += n * ops[b][k]
    valu = tot["v_mul"] + tot["v_other"] + tot["v_mov"] + tot["v_cndmask"] + tot["v_cmp"]
    print("%-10s %6s " % ("weighted", "") + " ".join("%9d" % tot[k] for k in keys) + " %8d" % valu)
    print("issue slots (mul x2, s_nop as 1): %d" % (valu + tot["v_mul"] + tot["s_nop"]))
    print("top ops:", ", ".join("%s=%d" % kv for kv in totops.most_common(24)))

if __name__ == "__main__":
    main()
