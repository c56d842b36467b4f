// Handle definitions shared by the API translation units.
#pragma once
#include "p2s_common.h"
#include <vector>

struct p2s_model_s {
    p2s_model_cfg cfg;
    p2s_weight_offsets offs;
    int device = 0;
    float *blob = nullptr;
    size_t n_floats = 0;
    float *ws = nullptr;      // per-chunk workspace, grown on demand
    int ws_chunk = 0;
    int max_chunk = 4096;     // queries per internal batch
    // profiling: HIP events recorded on the launch stream, no host synchronisation until collect
    bool profiling = false;
    std::vector<hipEvent_t> evpool;
    int ev_used = 0;
    struct Span { int stage, a, b; };
    std::vector<Span> spans;
    p2s_counters counters = {};
    // auxiliary stream: the data path (kNN, sub-sample) of chunk i+1, i+2 overlaps the encoders of chunk i
    hipStream_t aux = nullptr;     // serial MT19937 sub-sample stream: CU-masked to its own 2 CUs
    hipStream_t comp = nullptr;    // everything else of p2s_infer_shape: the remaining CUs
    bool overlap = true;
};

enum P2SStage { ST_CHAIN_STN = 0, ST_HEAD, ST_CHAIN_MAIN, ST_DECODER, ST_KNN, ST_SUB, ST_GRID };
int p2s_prof_mark(p2s_model_s *m, hipStream_t s);                 // event index or -1
void p2s_prof_span(p2s_model_s *m, int stage, int a, int b);
void p2s_prof_reset(p2s_model_s *m);
void p2s_prof_collect(p2s_model_s *m);                            // synchronises the last event

int p2s_model_reserve(p2s_model_s *m, int chunk);
int p2s_run_chunk(p2s_model_s *m, const float *patch, const float *sub, const float *query, const float *radius,
                  int C, float *logits_out, float *sdf_out, float *feat_local_out, float *feat_global_out,
                  hipStream_t s);
