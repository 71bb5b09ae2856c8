// tests/hostcheck/pipeline_driver.cpp — drives the batched pipeline through the C ABI ALONE: no Python, no torch, no HIP
// headers — exactly what a Rust / C host binding include/bsx.h would do.  Test harness only.
//
//   pipeline_driver <case file> [--rccl]
// --rccl: the all-gather of the multi-GPU path is RCCL called from the library (bsx_pipeline_set_rccl): the driver makes a
// one-rank communicator with bsx_rccl_get_unique_id / bsx_rccl_comm_init_rank, checks the collective end to end
// (bsx_pipeline_check_allgather) and runs the case with it set — what a Rust host does at world N, without any Python.
// The case file (written by tests/test_gpu_engine.py from a synthetic workload and the oracle's verdicts) holds the inputs of R
// header_range instances and the expected public outputs / statuses.  The driver creates a pipeline of E chunks, uploads,
// enqueues `steps` steps back to back WITHOUT joining, fetches the results and compares.  Exit code 0 = every byte equal.
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/bsx.h"

struct CaseHeader {
    uint32_t magic, J, B, V, R, E, hpr, steps, flags, chain_id_len;
    uint8_t chain_id[56];
};

static bool read_exact(FILE* f, void* p, size_t n) { return fread(p, 1, n, f) == n; }

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s <case file> [--rccl]\n", argv[0]); return 2; }
    const bool with_rccl = argc > 2 && strcmp(argv[2], "--rccl") == 0;
    if (bsx_prepare_process() < 0) { fprintf(stderr, "bsx_prepare_process failed\n"); return 3; }   // before the first HIP call
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror("open"); return 2; }
    CaseHeader h;
    if (!read_exact(f, &h, sizeof h) || h.magic != 0x42535850u) { fprintf(stderr, "bad case file\n"); return 2; }
    const size_t R = h.R, V = h.V, HPR = h.hpr;
    std::vector<bsx_header> headers(R * HPR);
    std::vector<bsx_shared_ctx> ranges(R);
    std::vector<uint64_t> latest(R);
    std::vector<bsx_validator> tv(R * V), rv(R * V);
    std::vector<uint8_t> want_out(R * 64);
    std::vector<uint32_t> want_rs(R), want_ss(R);
    if (!read_exact(f, headers.data(), headers.size() * sizeof(bsx_header)) || !read_exact(f, ranges.data(), R * sizeof(bsx_shared_ctx)) ||
        !read_exact(f, latest.data(), R * 8) || !read_exact(f, tv.data(), tv.size() * sizeof(bsx_validator)) ||
        !read_exact(f, rv.data(), rv.size() * sizeof(bsx_validator)) || !read_exact(f, want_out.data(), R * 64) ||
        !read_exact(f, want_rs.data(), R * 4) || !read_exact(f, want_ss.data(), R * 4)) {
        fprintf(stderr, "truncated case file\n");
        return 2;
    }
    fclose(f);

    bsx_ctx* ctx = nullptr;
    if (bsx_init(0, &ctx) != BSX_OK) { fprintf(stderr, "bsx_init: %s\n", bsx_last_error()); return 3; }
    bsx_pipeline_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.nb_map_jobs = h.J; cfg.batch_size = h.B; cfg.v_max = h.V; cfg.n_ranges = h.R; cfg.n_chunks = h.E;
    cfg.rank = 0; cfg.world = 1; cfg.flags = h.flags;
    cfg.chain_id_len = h.chain_id_len;
    memcpy(cfg.chain_id, h.chain_id, h.chain_id_len);
    bsx_pipeline* p = nullptr;
    if (bsx_pipeline_create(ctx, &cfg, &p) != BSX_OK) { fprintf(stderr, "bsx_pipeline_create: %s\n", bsx_last_error()); return 3; }
    void* comm = nullptr;
    if (with_rccl) {
        uint8_t id[128];
        if (bsx_rccl_get_unique_id(id) != BSX_OK || bsx_rccl_comm_init_rank(ctx, 1, id, 0, &comm) != BSX_OK) { fprintf(stderr, "rccl: %s\n", bsx_last_error()); return 4; }
        if (bsx_pipeline_set_rccl(p, comm) != BSX_OK || bsx_pipeline_check_allgather(p) != BSX_OK) { fprintf(stderr, "rccl: %s\n", bsx_last_error()); return 4; }
        fprintf(stderr, "rccl: ncclAllGather from the C tier verified on a one-rank communicator\n");
    }
    bsx_pipeline_inputs in;
    memset(&in, 0, sizeof in);
    in.headers = headers.data(); in.headers_per_range = HPR; in.ranges = ranges.data(); in.latest = latest.data();
    in.target_validators = tv.data(); in.trusted_validators = rv.data();
    if (bsx_pipeline_upload(p, &in) != BSX_OK) { fprintf(stderr, "bsx_pipeline_upload: %s\n", bsx_last_error()); return 3; }
    bsx_pipeline_autotune_result tune;
    if (bsx_pipeline_autotune(p, 2, &tune) != BSX_OK) { fprintf(stderr, "bsx_pipeline_autotune: %s\n", bsx_last_error()); return 3; }
    fprintf(stderr, "autotune: %u placements tried, %.3f -> %.3f ms per step, %u hardware queues\n", tune.n_trials, tune.initial_ms, tune.best_ms, tune.hw_queues);
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t s = 0; s < h.steps; s++)
        if (bsx_pipeline_step(p) != BSX_OK) { fprintf(stderr, "bsx_pipeline_step: %s\n", bsx_last_error()); return 3; }
    if (bsx_pipeline_join(p) != BSX_OK) { fprintf(stderr, "bsx_pipeline_join: %s\n", bsx_last_error()); return 3; }
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    std::vector<uint8_t> out(R * 64);
    std::vector<uint32_t> rs(R), ss(R);
    std::vector<bsx_commit_result> cr(R);
    bsx_pipeline_results res;
    memset(&res, 0, sizeof res);
    res.output64 = out.data(); res.range_status = rs.data(); res.skip_status = ss.data(); res.commit = cr.data();
    if (bsx_pipeline_get_results(p, &res) != BSX_OK) { fprintf(stderr, "bsx_pipeline_get_results: %s\n", bsx_last_error()); return 3; }
    int bad = 0;
    if (res.header_status || res.assemble_status) { fprintf(stderr, "device status words: header %u hint %u\n", res.header_status, res.assemble_status); bad++; }
    for (size_t r = 0; r < R; r++) {
        if (memcmp(&out[r * 64], &want_out[r * 64], 64) != 0) { fprintf(stderr, "range %zu: public output differs\n", r); bad++; }
        if ((rs[r] != 0) != (want_rs[r] != 0)) { fprintf(stderr, "range %zu: range status %u, expected %s\n", r, rs[r], want_rs[r] ? "failure" : "0"); bad++; }
        if (ss[r] != want_ss[r]) { fprintf(stderr, "range %zu: skip status %u, expected %u\n", r, ss[r], want_ss[r]); bad++; }
    }
    // the witness stays on the device: report where a prover would read it
    void* wptr = nullptr;
    uint64_t wbytes = 0;
    (void)bsx_pipeline_buffer(p, 0, BSX_PIPE_BUF_WITNESS_MAP, &wptr, &wbytes);
    if (with_rccl && bsx_pipeline_check_allgather(p) != BSX_OK) { fprintf(stderr, "rccl after the steps: %s\n", bsx_last_error()); bad++; }
    bsx_pipeline_destroy(p);
    if (comm && bsx_rccl_comm_destroy(comm) != BSX_OK) { fprintf(stderr, "rccl: %s\n", bsx_last_error()); bad++; }
    bsx_shutdown(ctx);
    printf("%s ranges=%zu steps=%u chunks=%u ms=%.3f witness_bytes_chunk0=%llu hw_queues=%u%s\n", bad ? "PIPELINE_DRIVER_MISMATCH" : "PIPELINE_DRIVER_OK", R, h.steps,
           h.E, ms, (unsigned long long)wbytes, tune.hw_queues, with_rccl ? " rccl=ok" : "");
    return bad ? 1 : 0;
}
