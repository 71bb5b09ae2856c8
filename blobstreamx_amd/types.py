"""POD layouts of include/bsx.h as numpy structured dtypes (data formats only, no compute).

Each dtype's itemsize is asserted against the C struct size the header documents; arrays of these
dtypes are passed to the C ABI as plain pointers (``arr.ctypes.data``).

Reference types mirrored (file:line under the reference tree):
  HEADER        tendermint Header as 14 encoded Merkle leaves (circuits/input.rs:250-261)
  DH_PROOF      InclusionProof<4, 34>  (circuits/input.rs:203-207, circuits/vars.rs:18-21)
  LB_PROOF      InclusionProof<4, 72>  (circuits/input.rs:208-217, circuits/vars.rs:22-25)
  SHARED_CTX    DataCommitmentSharedCtx (circuits/builder.rs:12-18)
  SUBCHAIN      MapReduceSubchainVariable (circuits/vars.rs:28-36)
  VALIDATOR     one validator slot of SkipOffchainInputs / StepOffchainInputs [UPSTREAM tendermintx]
"""
import numpy as np

HASH_SIZE = 32
PROTOBUF_HASH_SIZE_BYTES = 34       # circuits/consts.rs:4
PROTOBUF_BLOCK_ID_SIZE_BYTES = 72   # circuits/consts.rs:7
HEADER_PROOF_DEPTH = 4              # circuits/consts.rs:10
ENC_DATA_ROOT_TUPLE_SIZE_BYTES = 64  # circuits/consts.rs:18
BLOCK_HEIGHT_INDEX = 2              # circuits/consts.rs:21
LAST_BLOCK_ID_INDEX = 4             # circuits/consts.rs:22
DATA_HASH_INDEX = 6                 # circuits/consts.rs:23
VALIDATOR_MSG_MAX = 124
MAX_BATCH = 256

# bsx_status
OK, ERR_NO_DEVICE, ERR_HIP, ERR_BAD_ARG, ERR_RANGE_TOO_LONG, ERR_BAD_HEADER, ERR_ASSERT, ERR_BAD_SIGNATURE, \
    ERR_VOTING_POWER, ERR_UNSUPPORTED = range(10)
TUNE_MERKLE_WORKGROUPS = 1      # bsx_set_tuning key (bsx.h)
TUNE_HOST_GRAPHS = 2
STATUS_NAMES = ["OK", "ERR_NO_DEVICE", "ERR_HIP", "ERR_BAD_ARG", "ERR_RANGE_TOO_LONG", "ERR_BAD_HEADER", "ERR_ASSERT",
                "ERR_BAD_SIGNATURE", "ERR_VOTING_POWER", "ERR_UNSUPPORTED"]

A1_END_GTE_START = 1 << 0
A2_NB_BLOCKS_U32 = 1 << 1
A3_PREV_HEADER = 1 << 2
A4_DATA_HASH_PROOF = 1 << 3
A5_END_HEADER = 1 << 4
A6_BATCH_END = 1 << 5
A7_RANGE = 1 << 6
A8_REDUCE_LINK = 1 << 7
A9_FINAL = 1 << 8
A10_NEXT_HEADER = 1 << 9

HEADER = np.dtype([
    ("len", "u1", 14), ("_pad", "u1", 2),
    ("version", "u1", 24), ("chain_id", "u1", 52), ("height", "u1", 12), ("time", "u1", 20),
    ("last_block_id", "u1", 76), ("hash", "u1", (8, 36)), ("proposer", "u1", 24)])
DH_PROOF = np.dtype([("aunts", "u1", (4, 32)), ("leaf", "u1", 34)])
LB_PROOF = np.dtype([("aunts", "u1", (4, 32)), ("leaf", "u1", 72)])
SHARED_CTX = np.dtype([("start_block", "<u8"), ("end_block", "<u8"),
                       ("start_header_hash", "u1", 32), ("end_header_hash", "u1", 32)])
SUBCHAIN = np.dtype([("start_block", "<u8"), ("end_block", "<u8"), ("start_header", "u1", 32),
                     ("end_header", "u1", 32), ("data_merkle_root", "u1", 32), ("is_enabled", "<u4"),
                     ("assert_fail", "<u4"), ("first_bad_slot", "<u4"), ("_pad", "<u4")])
VALIDATOR = np.dtype([("pubkey", "u1", 32), ("signature", "u1", 64), ("message", "u1", 124),
                      ("message_len", "<u4"), ("voting_power", "<u8"), ("enabled", "u1"), ("is_signed", "u1"),
                      ("present_on_trusted", "u1"), ("_pad", "u1", 21)])
SKIP_EVAL = np.dtype([("overlap_power", "<u8"), ("start_total_power", "<u8"), ("signed_power", "<u8"),
                      ("target_total_power", "<u8"), ("valid", "<u4"), ("power_overflow", "<u4")])      # bsx_skip_eval, 40 B
COMMIT_RESULT = np.dtype([("validators_hash", "u1", 32), ("total_power", "<u8"), ("signed_power", "<u8"),
                          ("trusted_signed_power", "<u8"), ("n_enabled", "<u4"), ("n_signed", "<u4"),
                          ("n_bad_signature", "<u4"), ("first_bad_signature", "<u4"), ("n_bad_message", "<u4"),
                          ("two_thirds_ok", "<u4"), ("power_overflow", "<u4"), ("_pad", "<u4", 3)])
WITNESS_LAYOUT = np.dtype([("batch_size", "<u4"), ("n_bytes", "<u4"), ("n_words", "<u4"), ("n_bools", "<u4"),
                           ("compact_stride", "<u4"), ("off_words", "<u4"), ("off_bools", "<u4"), ("_pad", "<u4"),
                           ("n_elements", "<u8")])

COMMIT_FOLD = np.dtype([("root", "u1", 32), ("n_commits", "<u8"), ("n_ok", "<u8"), ("n_signatures_ok", "<u8"), ("first_index", "<u4"),
                        ("first_failing", "<u4"), ("_pad", "<u4", 16)])      # bsx_commit_fold, 128 B
assert COMMIT_FOLD.itemsize == 128
assert HEADER.itemsize == 512
assert DH_PROOF.itemsize == 162 and LB_PROOF.itemsize == 200
assert SHARED_CTX.itemsize == 80 and SUBCHAIN.itemsize == 128
assert VALIDATOR.itemsize == 256 and COMMIT_RESULT.itemsize == 96
assert WITNESS_LAYOUT.itemsize == 40
MANIFEST_ENTRY = np.dtype([("name", "S80"), ("reference", "S24"), ("kind", "<u4"), ("repeat", "<u4"), ("element_offset", "<u8"),
                           ("elements_per_record", "<u8"), ("record_stride", "<u8")])
assert MANIFEST_ENTRY.itemsize == 136
KIND_BYTES, KIND_U32, KIND_BOOL = 0, 1, 2

HEADER_FIELD_NAMES = ["version", "chain_id", "height", "time", "last_block_id"] + ["hash"] * 8 + ["proposer"]
HEADER_FIELD_CAP = [24, 52, 12, 20, 76] + [36] * 8 + [24]


def pack_header(fields):
    """14 encoded field byte strings -> one HEADER record (raises ValueError like BSX_ERR_BAD_HEADER)."""
    if len(fields) != 14:
        raise ValueError("a Tendermint header has 14 Merkle leaves")
    h = np.zeros((), dtype=HEADER)
    for i, f in enumerate(fields):
        cap = HEADER_FIELD_CAP[i]
        if len(f) > cap or (i != 4 and len(f) > 54):
            raise ValueError(f"header field {i} is {len(f)} bytes (capacity {cap})")
        h["len"][i] = len(f)
        dst = h["hash"][i - 5] if 5 <= i <= 12 else h[HEADER_FIELD_NAMES[i]]
        dst[:len(f)] = np.frombuffer(bytes(f), dtype=np.uint8)
    return h


def header_fields(h):
    """Inverse of pack_header: the 14 encoded fields of one HEADER record as bytes."""
    out = []
    for i in range(14):
        src = h["hash"][i - 5] if 5 <= i <= 12 else h[HEADER_FIELD_NAMES[i]]
        out.append(bytes(src[:int(h["len"][i])]))
    return out


def map_layout(batch_size):
    """Python twin of bsx_map_layout (include/bsx_layout.h) — used to size buffers."""
    B = batch_size
    n_bytes = 128 + (162 + 200) * B + 352 * B + 64 * B + 32 * B + 32 * (B - 1) * 2 + 96
    n_words = 20 + 4 * B
    n_bools = 1 + 9 * B + 6 + B + (B - 1) + 1
    return _layout(B, n_bytes, n_words, n_bools)


def reduce_layout():
    return _layout(0, 128, 4, 6)


def _layout(B, n_bytes, n_words, n_bools):
    a16 = lambda x: (x + 15) & ~15
    lay = np.zeros((), dtype=WITNESS_LAYOUT)
    lay["batch_size"], lay["n_bytes"], lay["n_words"], lay["n_bools"] = B, n_bytes, n_words, n_bools
    lay["off_words"] = a16(n_bytes)
    lay["off_bools"] = int(lay["off_words"]) + a16(4 * n_words)
    lay["compact_stride"] = int(lay["off_bools"]) + a16(n_bools)
    lay["n_elements"] = 8 * n_bytes + n_words + n_bools
    return lay


# ---- round 4: commit / skip / step units (include/bsx_layout.h; builder.skip / builder.step, header_range.rs:42-48, next_header.rs:32-36)
SECTION_MAP, SECTION_REDUCE, SECTION_COMMIT, SECTION_SKIP, SECTION_STEP = range(5)


def pow2_ceil(v):
    p = 1
    while p < v:
        p *= 2
    return p


def commit_layout(v_max):
    """Python twin of bsx_commit_layout: the COMMIT unit (one commit of v_max validator slots)."""
    V, P = v_max, pow2_ceil(v_max)
    n_bytes = 32 + 64 * V + 32 * V + 48 * V + 32 * P + 64 * (P - 1) + 32 + 220 * V
    return _layout(V, n_bytes, 4 * V + 6, 7 * V + 2 * P - 1 + 3)


SKIP_PROOF_CAPS = (52, 12, 36, 36)
STEP_PROOF_CAPS = (52, 12, 36, 76, 36, 36)


def skip_layout(v_max):
    """Python twin of bsx_skip_layout: the SKIP unit of CombinedSkipCircuit::define."""
    V, P = v_max, pow2_ceil(v_max)
    n_bytes = 96 + 48 * V + 32 * P + 64 * (P - 1) + 32 + 32 * V + sum(288 + c for c in SKIP_PROOF_CAPS)
    return _layout(V, n_bytes, 12 + 3 * V, 2 * V + 2 * P - 1 + 9)


def step_layout():
    """Python twin of bsx_step_layout: the STEP unit of CombinedStepCircuit::define."""
    return _layout(0, 96 + sum(288 + c for c in STEP_PROOF_CAPS) + 64, 10, 10)


def header_range_witness_elements(nb_map_jobs, batch_size, v_max):
    return (nb_map_jobs * int(map_layout(batch_size)["n_elements"]) + (nb_map_jobs - 1) * int(reduce_layout()["n_elements"])
            + int(commit_layout(v_max)["n_elements"]) + int(skip_layout(v_max)["n_elements"]))


def next_header_witness_elements(v_max):
    return int(commit_layout(v_max)["n_elements"]) + int(step_layout()["n_elements"])
