// grim_engine_internal.h -- what the streaming pipeline (grim_stream.cpp) uses of the engine (grim_engine.hip) below
// the C-ABI: batches with CAPACITY (device buffers + pinned staging allocated once, reloaded chunk after chunk).
#pragma once
#include <stdint.h>

#include "../../include/grim_hip.h"
#include "grim_layout.h"

struct EngineCaps {
  uint32_t subj;    // subject records (and result headers)
  uint64_t tok;     // u16 tokens
  uint32_t priors;  // prior matrices (the engine adds the all-ones matrix itself)
  uint64_t rows;    // output row pool
};

struct EngineHost {  // pinned host memory of a batch: inputs are written here, results land here
  grim_subject *subj;
  uint16_t *tok;
  double *priors;
  SmallRec *small;
  uint32_t *order_s, *order_m, *order_g;
  grim_subject_result *res;
  grim_row *rows;
};

struct EngineLoad {
  uint32_t n_subj;     // subject records [0, n_subj) are copied (records outside every list are never read)
  uint32_t n_priors;
  uint32_t n_small, n_medium, n_general;  // entries of order_s (+ small), order_m, order_g
  uint32_t n_tok_spans;                   // token ranges in use: [off, off+len) in u16 units
  const uint64_t *tok_span_off, *tok_span_len;
};

grim_batch *engine_batch_create(grim_ctx *ctx, const grim_graph *g, const grim_params *p, const EngineCaps *caps);
int engine_batch_reserve(grim_batch *b, const EngineCaps *caps);  // grow (contents are lost)
const EngineHost *engine_batch_host(grim_batch *b);
EngineCaps engine_batch_caps(const grim_batch *b);
uint32_t engine_small_stride(const grim_params *p);
// one subject's worst case in output rows (all four tables full)
uint64_t engine_rows_per_subject(const grim_params *p, uint32_t n_pops);
// H2D of the used parts of the staging area (asynchronous, on the context's stream) and a clean run state
int engine_batch_load(grim_batch *b, const EngineLoad *ld);
// D2H of result headers [res_lo, res_hi) and rows [0, rows_used) into the pinned landing area (rows) or `rows_dst`
// when given; waits for the copies
int engine_batch_fetch(grim_batch *b, uint32_t res_lo, uint32_t res_hi, grim_row *rows_dst);
uint64_t engine_bytes_moved(const grim_batch *b, int dir);  // 0 = H2D, 1 = D2H since creation
void engine_set_error(grim_ctx *ctx, const char *msg);
// hand a batch back to its context: the next engine_batch_create on that context reuses its buffers
void engine_batch_recycle(grim_batch *b);
