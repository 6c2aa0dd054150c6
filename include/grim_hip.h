/*
 * grim_hip.h -- C-ABI of libgrim_hip.so, the MI355X (gfx950) engine for the grim.impute hot path.
 *
 * The reference (nmdp-bioinformatics/py-graph-imputation) has no FFI for this path -- its only
 * native code is a 65-line Cython list helper (grim/imputation/cutils.pyx) -- so this ABI is the
 * build's own.  Each entry point names the reference interface it replaces (file:line under the
 * reference tree).  Plain pointers and sizes only; the library copies what it is given, owns
 * what it returns, and never calls back into the host language.
 *
 * Threading: one grim_ctx per GPU.  The block entry points (grim_batch_*) of a context are used by one host thread at
 * a time; a grim_stream runs its own threads on its context (one stream per context at a time).  Independent contexts
 * are independent.  grim_last_error returns a copy private to the calling thread.
 * Errors: functions return 0 on success, <0 on error (grim_last_error gives the text);
 * constructors return NULL on error.  Per-subject outcomes (miss / needs-fallback) are DATA in
 * grim_subject_result.status, never a call failure.
 */
#ifndef GRIM_HIP_H
#define GRIM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GRIM_MAXL 5        /* loci per haplotype (A,B,C,DQB1,DRB1)                         */
#define GRIM_ABITS 12      /* bits per locus in a 64-bit haplotype key: (allele_id+1)<<12*slot */
#define GRIM_KEY_GRAPH_ORDER 60 /* bit of a haplotype key in a result row: the haplotype is a graph node's NAME (alleles in
                                  loci_map index order), not a joined key (alleles sorted); matters to the phased writer only,
                                  and only under a loci_map that is not alphabetical */
#define GRIM_MAXPH 16      /* 2^(GRIM_MAXL-1) phases (impute.py:274-303)                     */
#define GRIM_MAXPOP 64     /* populations                                                    */
#define GRIM_TOPCAP 128    /* upper bound for max_haplotypes_number_in_phase (impute.py:436) */
#define GRIM_MAXLADDER 64  /* epsilon ladder entries (impute.py:1665-1687)                   */
#define GRIM_MAXROWS 8     /* Plan_B_Matrix rows                                             */

typedef struct grim_ctx grim_ctx;
typedef struct grim_graph grim_graph;
typedef struct grim_batch grim_batch;

/* ---- context ---------------------------------------------------------------------------- */
grim_ctx *grim_create(int device_id);
void grim_destroy(grim_ctx *ctx);
const char *grim_last_error(grim_ctx *ctx);
/* 1 if a HIP device is usable by this process, 0 otherwise (never raises). */
int grim_device_count(void);
/* How a stream's results come from HBM to pinned host memory on this context: the bit of the SDMA engine the downloads are
 * pinned to (hsa_amd_sdma_engine_id_t, > 1: csrc/grim_sdma.h -- uploads stay on engine 0x1, so both directions of the link
 * run at once), 0 = a copy kernel (GRIM_EXPORT=kernel, or no second engine on offer), -1 = hipMemcpyAsync
 * (GRIM_EXPORT=memcpy).  The reference has no counterpart (its results are Python objects). */
int grim_export_engine(grim_ctx *ctx);

/* ---- graph: replaces Graph.build_graph's in-memory product (networkx_graph.py:42-213) ----
 * Integer-encoded: node = (label bitmask over locus slots, <=5 allele ids); a node NAME lookup
 * (Vertices_attributes[name], networkx_graph.py:260,290) becomes a 64-bit key probe.
 * a_start/a_nbr : plan-A CSR partial node -> full nodes, INCLUDING the reference's row-start
 *                 quirks (networkx_graph.py:157-198); a_start has n_nodes+1 entries.
 * b_conn        : [n_nodes][GRIM_MAXL] connector index of (child node, added locus slot), or
 *                 0xFFFFFFFF (the "parentLabel+childName" pseudo nodes, networkx_graph.py:103-130)
 * b_start/b_nbr : connector -> parent nodes CSR (n_conn+1 entries, quirks included).
 * lab_start/lab_nodes : node ids grouped by label mask in id order (haps_by_label,
 *                 networkx_graph.py:215-236); lab_start has (1<<GRIM_MAXL)+1 entries. */
typedef struct {
  uint32_t n_nodes, n_pops, n_loci, full_mask;
  const uint64_t *node_key;  /* [n_nodes] */
  const uint8_t *node_mask;  /* [n_nodes] */
  const double *freq;        /* [n_nodes][n_pops] */
  const uint32_t *a_start;   /* [n_nodes+1] */
  const uint32_t *a_nbr;     /* [n_a_nbr] */
  uint64_t n_a_nbr;
  const uint32_t *b_conn;    /* [n_nodes*GRIM_MAXL] */
  const uint32_t *b_start;   /* [n_conn+1] */
  const uint32_t *b_nbr;     /* [n_b_nbr] */
  uint32_t n_conn;
  uint64_t n_b_nbr;
  const uint32_t *lab_start; /* [(1<<GRIM_MAXL)+1] */
  const uint32_t *lab_nodes; /* [n_nodes] */
  /* bit m set: a haplotype NAME over the loci of label mask m built from a subject's alleles never equals a graph name --
   * the subject's alleles come in sorted STRING order (gl2haps, impute.py:271), graph names in loci_map index order
   * (generate_neo4j_multi_hpf.py:59-68), and for this locus set the two orders differ (a loci_map that is not alphabetical,
   * e.g. the reference's own default A 1, B 3, C 2: every set that holds both B and C).  0: the two orders agree everywhere. */
  uint32_t label_order_bad;
  uint32_t reserved;
} grim_graph_desc;

grim_graph *grim_graph_upload(grim_ctx *ctx, const grim_graph_desc *desc);
void grim_graph_free(grim_graph *g);
uint64_t grim_graph_device_bytes(const grim_graph *g);

/* ---- graph, host side (C++, no GPU) ------------------------------------------------------
 * grim_hostgraph_load_csv: nodes.csv / top_links.csv / edges.csv -> the arrays above, i.e. all of
 *   Graph.build_graph (networkx_graph.py:42-213; argument order nodes, top links, all edges as at
 *   :42); alleles are interned into `dict` in file order.  NULL + message in err on failure.
 * grim_graphgen_csv: hpf.csv -> the four graph CSVs, i.e. generate_graph
 *   (graph_generation/generate_neo4j_multi_hpf.py:209-486); cutoff[p] = freq_trim_threshold /
 *   population count as computed at :251-262; locus_names/locus_index = the conf's loci_map. */
typedef struct grim_hostgraph grim_hostgraph;
typedef struct grim_dict grim_dict;     /* per locus slot: allele string <-> dense id (see the host helpers below) */
grim_hostgraph *grim_hostgraph_load_csv(grim_dict *dict, const char *full_loci, const char *nodes_csv, const char *top_links_csv,
                                        const char *edges_csv, char *err, uint64_t err_cap);
int grim_hostgraph_desc(const grim_hostgraph *h, grim_graph_desc *out);
void grim_hostgraph_free(grim_hostgraph *h);
int grim_graphgen_csv(const char *hpf_csv, const char *const *pops, const double *cutoff, uint32_t n_pops,
                      const char *const *locus_names, const uint32_t *locus_index, uint32_t n_locus_names, const char *nodes_csv,
                      const char *edges_csv, const char *top_links_csv, const char *info_csv, char *err, uint64_t err_cap);
/* grim_hostgraph_from_hpf: both of the above in one call -- hpf.csv -> loaded arrays, the generator's texts handed to the
 *   loader in memory (graph_freqs + Graph.build_graph, grim/grim.py:53-87, without the four files in between).  Each CSV
 *   path may be NULL; a path that is given is written exactly as grim_graphgen_csv writes it. */
grim_hostgraph *grim_hostgraph_from_hpf(grim_dict *dict, const char *full_loci, const char *hpf_csv, const char *const *pops,
                                        const double *cutoff, uint32_t n_pops, const char *const *locus_names,
                                        const uint32_t *locus_index, uint32_t n_locus_names, const char *nodes_csv,
                                        const char *edges_csv, const char *top_links_csv, const char *info_csv, char *err,
                                        uint64_t err_cap);

/* ---- run parameters: the conf keys of run_impute_def.py:63-129 that reach the hot path ------ */
typedef struct {
  double ladder[GRIM_MAXLADDER]; /* eps values tried in order (impute.py:1665-1673), host-computed */
  int32_t n_ladder;
  uint32_t top_n;          /* max_haplotypes_number_in_phase */
  uint64_t opt_threshold;  /* number_of_options_threshold    */
  uint32_t n_results;      /* number_of_results              */
  uint32_t n_pop_results;  /* number_of_pop_results          */
  uint8_t out_muug, out_haps, planb, em_mr;
  uint8_t em;              /* impute_file(em=True): the phased pass never falls back to Plan C (impute.py:1649) */
  uint8_t save_mode;       /* save_space_mode: open_option_ keeps the 10 largest entries of either operand before the cross
                              product (impute.py:1048-1059) */
  uint8_t eps_nonpositive; /* conf epsilon <= 0: call_comp_phase_prob never runs a pass and returns its "NaN" sentinel
                              (impute.py:1663-1665); the writers raise on it, so every subject that has phases ends as its
                              raw line in .problem (formatter rule; the ladder is empty) */
  uint8_t pop_rank[GRIM_MAXPOP]; /* rank of each population NAME in sorted order (impute.py:535) */
  double factor_missing_pow[GRIM_MAXL + 1]; /* factor_missing_data ** k for k = 0..5, evaluated by the
                                               host exactly as the reference does (impute.py:1168) */
  /* Plan_B_Matrix: row r has planb_nblk[r] blocks; block b is a locus-slot bitmask */
  uint8_t planb_rows;
  uint8_t planb_nblk[GRIM_MAXROWS];
  uint8_t planb_blk[GRIM_MAXROWS][GRIM_MAXL];
} grim_params;

/* ---- subjects: the tokenised form of one input line (impute.py:2022-2036, 105-118, 246-272) --
 * Position k (0..n_loci-1) is the k-th locus in the subject's sorted order; slot[k] is its locus
 * slot.  tokens[tok_off ...] holds, for k = 0.., side 0 then side 1, cnt[k][side] allele ids
 * (first occurrences only; wid[k][side] keeps the original '/'-list length that the
 * number_of_options_threshold test uses, impute.py:926-932). */
typedef struct {
  uint32_t tok_off;
  uint16_t prior_idx; /* index into the batch's prior matrices (impute.py:1956-1975) */
  uint8_t n_loci;
  uint8_t flags;      /* bit k: position k keeps its side in every phase (phase mask, impute.py:277-290) */
  uint8_t slot[GRIM_MAXL];
  uint8_t pad[3];
  uint16_t cnt[GRIM_MAXL][2];
  uint16_t wid[GRIM_MAXL][2];
  uint32_t reserved[2];
} grim_subject; /* 64 bytes */

typedef struct {
  uint32_t n_subjects;
  const grim_subject *subjects;
  const uint16_t *tokens;
  uint64_t n_tokens;
  const double *priors; /* [n_priors][n_pops][n_pops] */
  uint32_t n_priors;
} grim_batch_desc;

/* status values */
enum {
  GRIM_ST_OK = 0,         /* results present                                                    */
  GRIM_ST_MISS = 1,       /* nothing found on any plan  (-> .miss, impute.py:2065-2068)          */
  GRIM_ST_UNSUPPORTED = 2,/* needs a reference path this build does not run on device yet       */
  GRIM_ST_NOPHASE = 3     /* open_phases stayed empty after both rewrites (impute.py:1619-1629): the
                             reference writes nothing for MUUG and raises while writing the phased
                             output -> raw line in .problem when output_haplotypes is on             */
};

/* one output row: a/b are 64-bit haplotype keys (genotype and haplotype-pair tables); popa/popb are the populations of a
 * haplotype-pair row and NAME the pair of a population-table row (a/b repeat them there -- except in the one row that
 * serves as a one-population subject's genotype row and both its population rows: read popa/popb). */
typedef struct {
  uint64_t a, b;
  double prob;
  uint32_t popa, popb;
} grim_row; /* 32 bytes */

/* table ids inside grim_subject_result */
enum { GRIM_T_UMUG = 0, GRIM_T_UMUG_POPS = 1, GRIM_T_PMUG = 2, GRIM_T_PMUG_POPS = 3, GRIM_T_COUNT = 4 };

typedef struct {
  uint8_t status;
  uint8_t plan;           /* 'a', 'b' or 'c' as in impute.py:216,1638,1702 */
  uint8_t reason;         /* for GRIM_ST_UNSUPPORTED */
  uint8_t plan_phased;    /* plan the .pmug/.pmug.pops rows came from when it differs from `plan` (the phased pass
                             after a MUUG Plan C starts over, impute.py:1645-1654); 0 = same as `plan` */
  uint32_t n_pairs;       /* len(res_haps["Haps"])  (impute.py:2074-2078) */
  uint32_t n_genotypes;   /* len(res_muugs["Haps"]) (impute.py:2108-2112) */
  uint32_t row_off[GRIM_T_COUNT];
  uint32_t n_rows[GRIM_T_COUNT];
  double max_prob;
} grim_subject_result; /* 56 bytes */

/* grim_batch_upload copies subjects/tokens/priors to HBM and allocates result + scratch buffers.
 * grim_batch_run enqueues the kernels on the context's stream and waits; it may be called
 * repeatedly on the same resident batch (bench.py does).  Replaces the per-subject loop
 * impute_file -> impute_one -> comp_cand -> call_comp_phase_prob (impute.py:2019-2059,
 * 1584-1724) for every subject of the batch at once. */
grim_batch *grim_batch_upload(grim_ctx *ctx, const grim_graph *g, const grim_params *p,
                              const grim_batch_desc *d);
int grim_batch_run(grim_batch *b);
/* n complete synchronous runs back to back (what a caller's loop around grim_batch_run does, without the
 * caller's per-iteration overhead); stops at the first failing run and returns its code */
int grim_batch_run_repeat(grim_batch *b, uint32_t n);
/* Timing mode (also GRIM_TIMING=1 in the environment): every kernel of a run is bracketed by its own start/stop
 * hipEvents on the launch stream (hipExtLaunchKernelGGL).  Off by default: a synchronous 10k-subject run costs
 * 22 us without, 29 us with the events. */
int grim_batch_set_timing(grim_batch *b, int on);
/* device time of the last grim_batch_run in timing mode (0 otherwise):
 * which = 0 all kernels, 1 half-wave + one-wave + general kernel, 2 plan-B/C kernel, 3 half-wave kernel, 4 general plan-A
 *         kernel, 5 one-wave kernel, 6 the table kernels, 7 the half-wave kernel's row compaction (each from the kernels' own
 *         start/stop events);
 * which | 0x10 = the mean of that figure over all runs since timing was switched on */
double grim_batch_kernel_ms(const grim_batch *b, int which);
/* algorithmic byte counters of the last run: [0] probes, [1] CSR neighbour ids,
 * [2] frequency vectors gathered, [3] output rows */
int grim_batch_counters(const grim_batch *b, uint64_t out[4]);
uint32_t grim_batch_total_rows(const grim_batch *b);
/* copy results to host: res[n_subjects], rows[grim_batch_total_rows] */
int grim_batch_results(grim_batch *b, grim_subject_result *res, grim_row *rows);
void grim_batch_free(grim_batch *b);

/* ======================= host-side helpers (CPU, multi-threaded; no GPU needed) ====================
 * Allele dictionary, GL tokenizer and result formatter in C++ (csrc/grim_host.cpp).  They replace,
 * for whole files at a time, the reference's per-line Python: impute_file's line handling
 * (impute.py:2022-2036), clean_up_gl (:105-118), gl2haps (:246-272), the writers' text and the
 * .miss/.problem rules (:24-99, 2061-2118) including CPython's str(float). */
typedef struct grim_parsed grim_parsed; /* a tokenised block of input lines           */
typedef struct grim_text grim_text;     /* the six output texts                       */

grim_dict *grim_dict_create(uint32_t n_loci);
void grim_dict_free(grim_dict *d);
int grim_dict_set_locus(grim_dict *d, uint32_t slot, const char *locus_name);
int32_t grim_dict_intern(grim_dict *d, uint32_t slot, const char *allele); /* id, created if new; <0 = full */
int32_t grim_dict_find(const grim_dict *d, uint32_t slot, const char *allele);
const char *grim_dict_name(const grim_dict *d, uint32_t slot, uint32_t id);
uint32_t grim_dict_count(const grim_dict *d, uint32_t slot);

/* per-line outcome kinds; GRIM_K_UNSUPPORTED: more distinct alleles at one locus of one subject than a key field
 * holds (reported like a GRIM_ST_UNSUPPORTED subject, reason 5); GRIM_K_UNSUPPORTED_GL: a GL string that names a locus
 * twice or mixes loci inside one entry once each side's entries are sorted (reason 8) -- the reference checks no locus
 * and pairs the entries by index (gl2haps, grim/imputation/impute.py:246-272); such a subject is REPORTED, never
 * answered differently */
enum { GRIM_K_DEVICE = 0, GRIM_K_PROBLEM_ID = 1, GRIM_K_PROBLEM_RAW = 2, GRIM_K_MISS_NO_DEVICE = 3, GRIM_K_UNSUPPORTED = 4,
       GRIM_K_UNSUPPORTED_GL = 5 };

/* text: '\n'-separated input lines ("id,GL[,race1,race2]" or '%'-separated).  The dictionary is only READ:
 * alleles it does not know get ids from grim_dict_count(slot) upwards that are private to the subject that
 * brought them (grim_parsed_allele gives their text).  subjects[i].prior_idx = index of the line's
 * (race1, race2) pair in grim_parsed_race(); the caller supplies one prior matrix per pair in that order. */
grim_parsed *grim_tokenize(grim_dict *d, const char *text, uint64_t len, int planb, int n_threads);
void grim_parsed_free(grim_parsed *p);
uint32_t grim_parsed_lines(const grim_parsed *p);
uint32_t grim_parsed_subjects(const grim_parsed *p);
const grim_subject *grim_parsed_subject_array(const grim_parsed *p);
const uint16_t *grim_parsed_tokens(const grim_parsed *p, uint64_t *n_tokens);
const uint8_t *grim_parsed_kinds(const grim_parsed *p);      /* [lines]                      */
const int32_t *grim_parsed_dev_index(const grim_parsed *p);  /* [lines] subject index or -1  */
uint32_t grim_parsed_n_races(const grim_parsed *p);
const char *grim_parsed_race(const grim_parsed *p, uint32_t i, int which);
const char *grim_parsed_id(const grim_parsed *p, uint32_t line, uint32_t *len);
/* text (not NUL terminated) of allele `id` at locus slot `slot` as line `line` uses it; NULL if there is none */
const char *grim_parsed_allele(const grim_parsed *p, uint32_t line, uint32_t slot, uint32_t id, uint32_t *len);
/* host-language overrides: force a line's outcome kind; set a subject's `flags` = bitmask of positions
 * that must NOT switch sides when phases are enumerated (bin_imputation_in_file, impute.py:277-290) */
int grim_parsed_set_kind(grim_parsed *p, uint32_t line, uint8_t kind);
int grim_parsed_set_flags(grim_parsed *p, uint32_t line, uint8_t flags);

/* res/rows as returned by grim_batch_results for the subjects of `p`.  line_offset = global index
 * of the first line (multi-GPU shards); skip = optional [lines] mask of lines to leave out; subjects with
 * status GRIM_ST_UNSUPPORTED are left out of every text.
 * Texts: 0 .umug, 1 .umug.pops, 2 .pmug, 3 .pmug.pops, 4 .miss, 5 .problem. */
grim_text *grim_format(const grim_dict *d, const grim_parsed *p, const grim_params *prm, const char *const *pop_names,
                       uint32_t n_pops, const grim_subject_result *res, const grim_row *rows, uint64_t line_offset,
                       const uint8_t *skip, int n_threads);
const char *grim_text_get(const grim_text *t, int which, uint64_t *len);
void grim_text_free(grim_text *t);
int grim_format_double(double x, char *buf, int cap); /* CPython str(float) */

/* ---- prior matrices: calc_priority_matrix (impute.py:1844-1924) --------------------------------------------------
 * One P x P matrix per (race1, race2) pair of an input line: alpha/beta/gamma/delta/eta weights of the conf's
 * "priority", the UNK_priors base matrix ("MR" all ones, else identity) for lines without a known population,
 * population counts of pops_count_file (impute.py:205-212; NULL = all 1).  Every operation in the reference's
 * order: the matrices are bit-identical to numpy's. */
typedef struct {
  double alpha, eta, beta, gamma, delta;
  uint8_t unk_mr;
  const double *count_by_prob; /* [n_pops] or NULL */
} grim_prior_spec;
int grim_prior_matrix(const grim_prior_spec *spec, const char *const *pop_names, uint32_t n_pops, const char *race1,
                      const char *race2, double *out /* [n_pops * n_pops] */);

/* ======================= streaming pipeline: impute_file's loop (impute.py:2019-2144) =================================
 * Input text goes in as it comes (any split, lines may straddle calls); the library cuts it into chunks of
 * chunk_lines lines and runs, per chunk and overlapped between chunks:
 *     tokenizer threads -> pinned staging -> H2D -> kernels -> D2H -> formatter threads -> ordered output
 * on `n_threads` host threads plus one device thread.  Output: the six texts, appended to the files named in
 * out_path (parallel pwrite at offsets known from the chunk order) or kept in memory; optionally the per-subject
 * stdout lines of impute_file (text 6); optionally the raw result records of every chunk (grim_stream_next_records).
 * The row pool of a chunk is bounded; a chunk that overflows it is run again -- with twice the pool (up to every line's
 * worst case) when rows_per_chunk was left at 0, in halves, down to single subjects, when the caller fixed it -- so no
 * configuration needs a worst-case allocation up front.
 * One stream per grim_ctx at a time; the calling thread may block in grim_stream_write while `depth` chunks are in
 * flight. */
typedef struct grim_stream grim_stream;
typedef struct {
  uint32_t chunk_lines;     /* lines per device batch; 0 = 131072 */
  uint32_t depth;           /* chunks in flight; 0 = 4 */
  int32_t n_threads;        /* tokenizer / formatter threads; 0 = grim_default_threads() */
  uint64_t line_offset;     /* global index of the first line (multi-GPU shards keep the reference's line numbers) */
  uint64_t rows_per_chunk;  /* row pool of one chunk; 0 = 32 rows per line to begin with, growing on demand; a value fixes it (never less than one subject's worst case) */
  uint8_t want_text;        /* format the six output texts */
  uint8_t want_log;         /* also text 6: the lines impute_file prints per subject */
  uint8_t want_records;     /* hand every chunk's records to grim_stream_next_records (the caller must drain them) */
  uint8_t timing;           /* per-kernel HIP events (grim_batch_set_timing) */
  uint8_t rows_exact;       /* take rows_per_chunk as it is (no floor): the caller knows that one subject's rows fit */
  uint8_t placed;           /* the out_path files are shared with other writers (the ranks of grim/shard.py): opened without
                             * truncation, and nothing is written until the caller gives a segment its base offsets
                             * (grim_stream_segment_wait / grim_stream_segment_place) */
  const char *out_path[6];  /* per text: file to create and fill, or NULL = keep in memory (grim_stream_text) */
  /* bin_imputation_in_file phase masks (impute.py:2001-2005, 2030-2032): n_masks ids (NUL-terminated, back to back
   * in mask_ids) and per id the bitmask of positions that keep their side; an id missing from the table sends its
   * line to .problem.  NULL = no masks. */
  const char *mask_ids;
  const uint8_t *mask_fixed;
  uint32_t n_masks;
} grim_stream_opts;

typedef struct {
  uint64_t lines, subjects, chunks, reruns;          /* reruns: sub-batches run again after a row-pool overflow */
  uint64_t unsupported;                              /* subjects left out (grim_stream_unsupported) */
  double wall_s;                                     /* open -> finish */
  double tokenize_cpu_s, format_cpu_s, write_cpu_s;  /* summed over the worker threads */
  double device_s;                                   /* device thread busy: copies + kernels + waits */
  double kernel_ms[7];                               /* sums of grim_batch_kernel_ms(which) over the chunks (timing mode) */
  uint64_t counters[4];                              /* sums of grim_batch_counters */
  uint64_t text_bytes[7];
  uint64_t bytes_h2d, bytes_d2h;
} grim_stream_stats;

grim_stream *grim_stream_open(grim_ctx *ctx, const grim_graph *g, const grim_dict *dict, const grim_params *prm,
                              const grim_prior_spec *priors, const char *const *pop_names, uint32_t n_pops,
                              const grim_stream_opts *opts);
int grim_stream_write(grim_stream *s, const char *text, uint64_t len);
/* The same, but the caller LENDS the bytes instead of having them copied: they must stay valid and unchanged until
 * grim_stream_finish has returned (or the stream is freed).  Chunks that begin inside the buffer read their lines where
 * they lie ('\n' line ends only).  For a caller that holds its input in memory anyway (a mapped file, a buffer it owns):
 * the reference's counterpart is the file object impute_file reads from (impute.py:2019-2027). */
int grim_stream_write_borrowed(grim_stream *s, const char *text, uint64_t len);
/* the same with universal newlines, as Python's open(): "\r\n" and "\r" end a line too (a "\r\n" may straddle two calls) */
int grim_stream_write_text(grim_stream *s, const char *text, uint64_t len);
/* reads the file in blocks and feeds it through grim_stream_write_text */
int grim_stream_write_file(grim_stream *s, const char *path);
/* Input segments -- what a rank of grim/shard.py feeds: byte ranges of ONE file that are not adjacent (the chunks it pulled
 * from the job's counter; scripts/runfile_mp.py:109-148 gives every worker ONE contiguous range instead).  grim_stream_segment
 * ends the current segment -- the line being filled is complete, with or without its newline -- and starts the next one at
 * global line index next_line_offset.  After grim_stream_finish, grim_stream_segment_end(k) gives the cumulative bytes of
 * the seven texts up to the end of segment k (segment 0 starts at the stream's line_offset), i.e. where segment k's part of
 * every output file ends. */
int grim_stream_segment(grim_stream *s, uint64_t next_line_offset);
uint32_t grim_stream_n_segments(const grim_stream *s);
int grim_stream_segment_end(const grim_stream *s, uint32_t k, uint64_t out[7]);
/* Placed output (opts.placed) -- how the ranks of a sharded job write ONE set of output files without a merge copy (what
 * scripts/runfile_mp.py:143-148 leaves to `cat`): grim_stream_segment_wait blocks until every chunk of the closed segment k
 * is formatted and gives the bytes of its piece of each text; the caller adds up the pieces before it (other ranks') and
 * grim_stream_segment_place writes segment k's buffers at those offsets of the shared files (returns when written). */
int grim_stream_segment_wait(grim_stream *s, uint32_t k, uint64_t sizes[7]);
int grim_stream_segment_place(grim_stream *s, uint32_t k, const uint64_t base[6]);
/* byte offsets of every chunk_lines-th line start of a file, the file size last (universal newlines, as
 * grim_stream_write_text; what scripts/runfile_mp.py:113-124 does with `split -l`, without writing the pieces): the number of
 * entries or -1; *out is the library's, released with grim_free */
int64_t grim_chunk_offsets(const char *path, uint32_t chunk_lines, uint64_t **out);
void grim_free(void *p);
/* worker threads a stream starts when opts.n_threads is 0: cores of this process's affinity mask / ranks on this host
 * (LOCAL_WORLD_SIZE, else WORLD_SIZE), at most 32 */
uint32_t grim_default_threads(void);
/* end of input: processes the last partial chunk and waits until every chunk is written; 0 or <0 (grim_stream_error) */
int grim_stream_finish(grim_stream *s);
const char *grim_stream_error(const grim_stream *s);
/* in-memory texts after grim_stream_finish (which: 0..5 as grim_format, 6 = log) */
const char *grim_stream_text(grim_stream *s, int which, uint64_t *len);
int grim_stream_get_stats(const grim_stream *s, grim_stream_stats *out);
/* subjects the device could not resolve (GRIM_ST_UNSUPPORTED / GRIM_K_UNSUPPORTED): global line index, reason, id */
uint64_t grim_stream_n_unsupported(const grim_stream *s);
int grim_stream_unsupported(const grim_stream *s, uint64_t k, uint64_t *line, uint32_t *reason, const char **id, uint32_t *id_len);
/* records mode: blocks until the next chunk (in input order) is done; 1 = a chunk, 0 = end of stream, <0 = error.
 * kinds[n_lines]; the subject of line j (kind GRIM_K_DEVICE) is res[j]; its rows are rows[res[j].row_off[t] ...].
 * The pointers stay valid until grim_stream_release_records; the chunk's buffers are not reused before that, so a
 * caller that never releases stalls the stream after `depth` chunks. */
typedef struct {
  uint64_t first_line;
  uint32_t n_lines;
  const uint8_t *kinds;
  const grim_subject_result *res;
  const grim_row *rows;
  void *chunk;
} grim_stream_records;
int grim_stream_next_records(grim_stream *s, grim_stream_records *out);
int grim_stream_release_records(grim_stream *s, grim_stream_records *rec);
void grim_stream_free(grim_stream *s);

#ifdef __cplusplus
}
#endif
#endif
