// kernels.h — the kernel launchers of libbsx.so (kernels_*.hip), as seen by the host-side translation units
// (api.hip, api_poseidon.hip, pipeline.hip).  Every launcher only enqueues on the given stream and returns hipGetLastError().
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/bsx.h"

extern "C" {
hipError_t bsxk_header_merkle(hipStream_t, const bsx_header*, uint64_t, uint8_t*, uint8_t*, uint8_t*, uint8_t*, uint32_t*, uint32_t, uint32_t);
hipError_t bsxk_zero_paths(hipStream_t, uint8_t*);
hipError_t bsxk_assemble_inputs(hipStream_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, const bsx_shared_ctx*,
                                const uint64_t*, const bsx_header*, uint64_t, uint64_t, const uint8_t*, const uint8_t*, const uint8_t*,
                                uint8_t*, uint32_t*, const uint8_t*, const uint8_t*);
hipError_t bsxk_prove_subchain(hipStream_t, uint32_t, uint32_t, uint32_t, const bsx_shared_ctx*, uint8_t*, bsx_subchain*, uint32_t);
hipError_t bsxk_reduce(hipStream_t, uint32_t, uint32_t, const bsx_subchain*, uint64_t, uint64_t, bsx_subchain*, uint8_t*);
hipError_t bsxk_finalize(hipStream_t, uint32_t, uint32_t, uint32_t, const bsx_shared_ctx*, const bsx_subchain*, const uint8_t*,
                         uint8_t*, uint32_t*);
hipError_t bsxk_expand_witness(hipStream_t, const bsx_witness_layout*, uint32_t, const uint8_t*, uint64_t*);
hipError_t bsxk_sha512_challenge(hipStream_t, const bsx_validator*, uint64_t, uint8_t*, uint8_t*);
hipError_t bsxk_ed25519_verify(hipStream_t, const bsx_validator*, const uint8_t*, uint64_t, uint8_t*);
uint64_t bsxk_keytable_bytes(uint32_t);
hipError_t bsxk_ed25519_keytable(hipStream_t, const bsx_validator*, uint32_t, uint8_t*);
hipError_t bsxk_ed25519_verify_keyed(hipStream_t, const bsx_validator*, const uint8_t*, uint64_t, uint32_t, const uint8_t*, uint32_t, const uint8_t*, uint8_t*, void*, const void*);
// sentinel for bsxk_ed25519_verify_keyed's last argument: no decoded R, and (with a scratch) prefer the form with the least total work
#ifndef BSXK_ED_THROUGHPUT
#define BSXK_ED_THROUGHPUT (reinterpret_cast<const void*>(static_cast<uintptr_t>(1)))
#endif
uint64_t bsxk_ed25519_rdec_bytes(uint64_t);
hipError_t bsxk_ed25519_decode_r(hipStream_t, const bsx_validator*, uint64_t, void*);
hipError_t bsxk_ed25519_btable(hipStream_t, uint8_t*);
uint64_t bsxk_ed25519_btable_bytes();
uint64_t bsxk_ed25519_scratch_bytes(uint64_t);
hipError_t bsxk_commit_tally(hipStream_t, const bsx_validator*, uint32_t, uint32_t, const uint8_t*, const uint8_t*, bsx_commit_result*);
hipError_t bsxk_skip_check(hipStream_t, uint32_t, uint32_t, const bsx_shared_ctx*, const bsx_header*, uint64_t, const uint8_t*,
                           const bsx_validator*, const bsx_validator*, const uint8_t*, bsx_commit_result*, const bsx_commit_result*,
                           uint32_t*, uint8_t*, const uint32_t*, const uint8_t*, uint32_t);
hipError_t bsxk_encode_tuple(hipStream_t, const uint8_t*, uint64_t, uint8_t*);
hipError_t bsxk_data_commitment(hipStream_t, const uint8_t*, uint32_t, uint64_t, uint64_t, uint8_t*, uint32_t*);
hipError_t bsxk_fill_end_hash(hipStream_t, uint32_t, bsx_shared_ctx*, const uint8_t*, uint64_t, const uint32_t*, uint8_t*, uint8_t*, uint64_t);
int bsxk_tally_vmax(void);
hipError_t bsxk_skip_eval(hipStream_t, const bsx_validator*, const bsx_validator*, uint32_t, uint32_t, bsx_skip_eval*);
uint64_t bsxk_commit_fold_scratch_bytes(uint32_t);
hipError_t bsxk_commit_fold(hipStream_t, const bsx_commit_result*, uint32_t, uint32_t, void*, bsx_commit_fold*);
}

extern "C" {
hipError_t bsxk_poseidon_permute(hipStream_t, const uint64_t*, uint64_t, uint64_t*);
hipError_t bsxk_leaf_hashes(hipStream_t, const bsx_witness_layout*, uint32_t, const uint8_t*, const uint64_t*, uint32_t, uint32_t, int,
                            uint64_t, uint64_t*);
hipError_t bsxk_merkle_caps(hipStream_t, uint64_t*, uint32_t, uint64_t, uint32_t, uint32_t);
hipError_t bsxk_merkle_one_level(hipStream_t, uint64_t*, uint64_t);
}
