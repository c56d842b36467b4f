// Handle definitions shared by the API translation units.
#pragma once
#include "p2s_common.h"

struct p2s_model_s {
    p2s_model_cfg cfg;
    p2s_weight_offsets offs;
    int device = 0;
    float *blob = nullptr;
    size_t n_floats = 0;
    float *ws = nullptr;      // per-chunk workspace, grown on demand
    int ws_chunk = 0;
    int max_chunk = 4096;     // queries per internal batch
    bool profiling = false;
    bool prof_pending = false;
    hipEvent_t ev[8] = {};
    p2s_counters counters = {};
};

int p2s_model_reserve(p2s_model_s *m, int chunk);
int p2s_run_chunk(p2s_model_s *m, const float *patch, const float *sub, const float *query, const float *radius,
                  int C, float *logits_out, float *sdf_out, float *feat_local_out, float *feat_global_out,
                  hipStream_t s);
void p2s_collect_profile(p2s_model_s *m);
