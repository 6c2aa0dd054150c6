// grim_engine_internal.h -- what the streaming pipeline (grim_stream.cpp) uses of the engine (grim_engine.hip) below
// the C-ABI: batches whose device and pinned-host arenas are allocated once and laid out again for every chunk, so that a
// chunk's whole input goes up in ONE copy and its results come back in ONE copy.
#pragma once
#include <stdint.h>

#include "../../include/grim_hip.h"
#include "grim_layout.h"
#include "grim_tok.h"

struct EnginePlan {   // what the next load holds at most
  uint32_t n_subj;    // subject records (= result headers)
  uint64_t tok_cap;   // u16 tokens
  uint64_t text_cap;  // bytes of chunk text for the device tokenizer (0: the batch never runs it)
};

struct EngineHost {  // pinned host memory of a batch: inputs are written here, results land here
  grim_subject *subj;
  uint16_t *tok;
  SmallRec *small;
  uint32_t *order_s, *order_m, *order_g;
  grim_subject_result *res;  // valid after engine_batch_fetch
  grim_row *rows;
  LineRec *lines;            // device tokenizer (EnginePlan.text_cap != 0): one record per line ...
  uint8_t *text;             // ... and the chunk's text
};

struct EngineLoad {
  uint32_t n_subj;                        // subject records [0, n_subj) (records outside every list are never read)
  uint32_t n_small, n_medium, n_general;  // entries of order_s (+ small), order_m, order_g
  uint64_t tok_used;                      // tokens [0, tok_used) are copied
  uint32_t n_priors;                      // prior matrices [n_priors][P*P] at `priors` (host memory, copied when the
  const double *priors;                   // count differs from what the batch holds; the set only ever grows)
  // device tokenizer: lines [dev_lo, dev_hi) of the n_subj line records are tokenised on the device (0, 0: none);
  // text_bytes of chunk text go up with them
  uint32_t dev_lo, dev_hi;
  uint64_t text_bytes;
};

// row_limit: rows one run may produce (the row pool); plan: first layout
grim_batch *engine_batch_create(grim_ctx *ctx, const grim_graph *g, const grim_params *p, uint64_t row_limit, const EnginePlan *plan);
// lay the arenas out for the next load (grows them when needed); engine_batch_host pointers change
int engine_batch_plan(grim_batch *b, const EnginePlan *plan);
const EngineHost *engine_batch_host(grim_batch *b);
uint32_t engine_small_stride(const grim_params *p);
// one subject's worst case in output rows (all four tables full)
uint64_t engine_rows_per_subject(const grim_params *p, uint32_t n_pops);
// one H2D copy of the input arena as far as it is used (asynchronous, on the context's stream), clean run state included
int engine_batch_load(grim_batch *b, const EngineLoad *ld);
// D2H of result headers [res_lo, res_hi) and rows [0, rows_used) into the pinned landing area (rows: or `rows_dst`
// when given); waits for the copies
int engine_batch_fetch(grim_batch *b, uint32_t res_lo, uint32_t res_hi, grim_row *rows_dst);
uint64_t engine_bytes_moved(const grim_batch *b, int dir);  // 0 = H2D, 1 = D2H since the library was loaded
void engine_set_error(grim_ctx *ctx, const char *msg);
// hand a batch back to its context: the next engine_batch_create on that context reuses its arenas
void engine_batch_recycle(grim_batch *b);
// after a run that returned -2: 1 = the pair pool ran out and the next load will make it big enough (run the same subjects
// again), 0 = split the batch
int engine_batch_fetch_async(grim_batch *b);
// ... in two halves: queue the copy (an event behind it) / wait for that event, possibly on another thread
int engine_batch_fetch_issue(grim_batch *b);
int engine_batch_fetch_wait(grim_batch *b);
// a run in two halves (grim_batch_run = enqueue + wait): stage 1 is launched behind whatever the context's stream holds and
// the call returns; engine_batch_wait (any thread, after enqueue returned) waits for it, runs stage 2 when the run state
// asks for it and returns grim_batch_run's code (0, -1, -2 = a pool overflowed)
int engine_batch_upload_done(grim_batch *b);  // the H2D copy of the last load is over (never blocks)
int engine_batch_enqueue(grim_batch *b);
int engine_batch_wait(grim_batch *b);
// device form of the allele dictionary (grim_tok.h), built once per stream from the frozen snapshot the host tokenizer
// reads; batches that run the device tokenizer are given it with engine_batch_set_dict
struct DictSnap;
typedef struct grim_devdict grim_devdict;
grim_devdict *engine_devdict_create(grim_ctx *ctx, const DictSnap *snap);
void engine_devdict_free(grim_devdict *d);
void engine_batch_set_dict(grim_batch *b, const grim_devdict *d);
// lines the device tokenizer of the last run handed back to the host (status GRIM_ST_UNSUPPORTED, reason 7)
uint32_t engine_batch_irregular(const grim_batch *b);
// DevGraph::order_bad of an uploaded graph (non-zero: a loci_map that is not alphabetical; general kernel only)
uint32_t engine_graph_order_bad(const grim_graph *g);
uint64_t engine_batch_pool_want(const grim_batch *b);
void engine_batch_hint_pool(grim_batch *b, uint64_t records);
int engine_batch_grow_pool(grim_batch *b, uint64_t max_records);
int engine_batch_grow_rows(grim_batch *b, uint64_t max_rows);
uint64_t engine_batch_row_limit(const grim_batch *b);
void engine_batch_hint_rows(grim_batch *b, uint64_t rows);
