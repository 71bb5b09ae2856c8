"""bench.py's shared helpers: device ceilings (bsx_calibrate), VALU-issue accounting, PMC traffic lookup, host description.
Nothing here touches oracle/ (only bench_legs/cpu.py and the `cpu_baseline` parts of the stress / commitment legs do)."""
import csv
import ctypes as C
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
# Field operations of ONE fixed-key Ed25519 verification (ed25519.h ed25519_verify_keyed_core: affine tables, 22 radix-4096
# digits of h for the key, 16 radix-65536 digits of s for B): 38 mixed additions (3 + 4 mul each, the last one 3 + 3), no
# doubling + encoding.  With the batch-inversion scratch (k_ed25519_finish) the encoding costs 5 multiplications per
# signature plus one inversion (254 sq + 11 mul) per 8 / 16 / 32 signatures — counted at 16.
# Round 5: a resident validator set's tables may hold 16-bit digits (BSX_COMMITS_KEYTABLE_WIDE): 16 + 16 = 32 additions.
def fe_mul_per_verify(kt_bits=12):
    adds = (253 + kt_bits) // kt_bits + 16
    return adds * 7 - 1 + 5 + 11 / 16
FE_MUL_PER_VERIFY = fe_mul_per_verify(12)
FE_SQ_PER_VERIFY = 254 / 16
# Goldilocks multiplications of one Poseidon permutation that NO formulation can avoid: the x^7 S-boxes (4 multiplications
# each: x2, x3 = x2*x, x4 = x2*x2, x7 = x4*x3) of 8 full rounds x 12 lanes + 22 partial rounds x 1 lane.  The MDS layers are
# multiplications by small constants (shifts/adds here) and are NOT counted: an upper-bound style ceiling, never below truth.
GL_MUL_PER_PERMUTATION = 4 * (8 * 12 + 22)

def log(msg):
    """progress on stderr (stdout carries the DETAIL line and the compact line)"""
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def host_threads():
    """Threads for the CPU legs and what the box really grants: the GPU boxes show 256 logical CPUs but run the container
    under a cgroup CPU quota (cpu.max 1600000/100000 = 16 CPUs): 256 threads then thrash the quota (152 k Ed25519 verifies/s
    vs 278 k at 32 threads, tools/cpu_probe.py).  -> (threads to use, description)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = int(q) / int(per)
    except Exception:
        pass
    model = "unknown CPU"
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except Exception:
        pass
    if quota and quota < n:
        t = max(1, min(n, int(round(2 * quota))))
        return t, f"{t} threads of {model} (cgroup CPU quota {quota:g} CPUs of {n} logical)"
    return n, f"{n} threads of {model} (no CPU quota)"


# ---------------------------------------------------------------------------------------------------------------- ceilings
class Calibration(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("valu_add_u32_lane_ops_per_s", "valu_mad_u64_u32_lane_ops_per_s", "valu_alignbit_lane_ops_per_s",
                                          "sha256_compress_per_s", "sha512_compress_per_s", "fe25519_mul_per_s", "fe25519_sq_per_s",
                                          "goldilocks_mul_per_s", "hbm_store_bytes_per_s")] + [("compute_units", C.c_uint32), ("clock_mhz", C.c_uint32)]


def calibrate(dev):
    """bsx_calibrate on this device: the ALU / store ceilings every ALU-bound `roofline.peak` below is priced against."""
    from blobstreamx_amd import _lib
    c = Calibration()
    _lib.check(_lib.lib().bsx_calibrate(_lib.context(dev.index or 0), C.byref(c)))
    d = {k: getattr(c, k) for k, _ in Calibration._fields_}
    d["source"] = "bsx_calibrate in this process (library's own device functions alone at 8 waves/SIMD; csrc/calibrate.hip)"
    return d


def keyed_verify_peak(cal, kt_bits=12):
    """Ed25519 verifications/s if only the kernel's field multiplications and squarings cost time, at the measured rates."""
    return 1.0 / (fe_mul_per_verify(kt_bits) / cal["fe25519_mul_per_s"] + FE_SQ_PER_VERIFY / cal["fe25519_sq_per_s"])


def valu_insts(kernel_substr):
    """Wave-level VALU instructions per unit of work of a kernel, from the committed SQ counter summaries (profiles/
    valu_insts.json, written by tools/valu_insts.py from `rocprofv3 --pmc SQ_INSTS_VALU` passes).  None if not profiled."""
    path = os.path.join(ROOT, "profiles", "valu_insts.json")
    if not os.path.exists(path):
        return None
    for k, v in json.load(open(path)).get("kernels", {}).items():
        if kernel_substr in k:
            return v
    return None


def valu_issue(cal, kernel_substr, units, seconds):
    """valu_issue_frac = SQ_INSTS_VALU (wave instructions, scaled to this launch) / (time x the measured v_add_u32 wave-issue
    rate of the device).  Independent of any body micro-benchmark: how much of the VALU issue bandwidth the kernel used."""
    v = valu_insts(kernel_substr)
    if not v or seconds <= 0:
        return None
    insts = v["wave_valu_insts_per_unit"] * units
    rate = cal["valu_add_u32_lane_ops_per_s"] / 64.0
    return {"valu_issue_frac": insts / (seconds * rate), "wave_valu_insts_per_unit": v["wave_valu_insts_per_unit"], "unit": v["unit"],
            "wave_issue_rate_per_s": rate, "counter_source": v.get("source")}


def pmc_traffic(n_units, match):
    """HBM bytes of one k_expand_witness launch of `n_units` units from the committed rocprofv3 PMC passes: the newest
    profiles/*pmc_hbm_traffic*.csv whose .meta.json (written by the profiling script) matches `match` — {"batch": B} for the map-job
    section of a header_range_{32 B}, {"layout": "commit", "v": V} for mode S's COMMIT units — scaled per unit.  (None, None) when no
    profile of that shape is committed."""
    import glob
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*pmc_hbm_traffic*.meta.json")), reverse=True)      # r5 before r4 ...
    for meta in cands:
        m = json.load(open(meta))
        m.setdefault("layout", "map")
        want = dict({"layout": "map"}, **match)
        if any(m.get(k) != v for k, v in want.items()):
            continue
        path = meta[:-len(".meta.json")] + ".csv"
        if not os.path.exists(path):
            continue
        kb = {}
        for r in csv.DictReader(open(path)):
            if "k_expand_witness" in r["kernel"] and r["counter"] in ("FETCH_SIZE", "WRITE_SIZE"):
                kb[r["counter"]] = max(kb.get(r["counter"], 0), int(r["per_launch_max"]))
        if len(kb) == 2:
            # rocprofv3 units are KB; FETCH_SIZE not doubled: the kernel reads its source with dword loads (guide: HBM section)
            per_launch = int(m.get("jobs_per_launch") or m.get("units_per_launch"))
            return (kb["FETCH_SIZE"] + kb["WRITE_SIZE"]) * 1024 / per_launch * n_units, os.path.basename(path)
    return None, None


def memory_partition_mode():
    for cmd in (["rocm-smi", "--showmemorypartition", "--showcomputepartition"], ["amd-smi", "partition"]):
        try:
            o = subprocess.run(cmd, capture_output=True, text=True, timeout=20)
            lines = [ln.strip() for ln in o.stdout.splitlines() if "artition" in ln and ":" in ln]
            if lines:
                return "; ".join(lines[:4])
        except Exception:
            pass
    return None
