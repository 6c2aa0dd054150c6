// grim_stream.cpp -- the streaming pipeline behind grim_stream_* (include/grim_hip.h): impute_file's loop
// (impute.py:2019-2144: read a line, parse, impute, write) as a chunked, overlapped pipeline
//
//     caller's bytes -> chunks of whole lines -> [worker threads] tokenize ranges of the chunk straight into the
//     chunk's pinned staging area -> [device thread] H2D, kernels, D2H into the pinned landing area ->
//     [worker threads] format ranges -> commit in input order (file offsets) -> [worker threads] pwrite
//
// `depth` chunks are in flight, each owning one capacity batch of the engine (grim_engine_internal.h), so the
// tokenizer of chunk k+1, the kernels of chunk k and the formatter of chunk k-1 run at the same time.  Subject number
// inside a chunk = line number inside the chunk (no compaction pass); tokens of a range go to the range's own slab of
// the staging area (one H2D copy per range); a chunk whose results overflow its bounded row pool is split and run
// again in halves.  No GPU code here: the device work goes through the engine.
#include <errno.h>
#include <execinfo.h>
#include <fcntl.h>
#include <signal.h>
#include <sched.h>
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <sys/types.h>
#include <unistd.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/grim_hip.h"
#include "grim_engine_internal.h"
#include "grim_host_internal.h"

static thread_local uint64_t tl_dbg[8];  // what copy_pieces was doing (read by the GRIM_DEBUG_SEGV handler)

namespace {

using Clock = std::chrono::steady_clock;
static inline double secs(Clock::time_point a, Clock::time_point b) { return std::chrono::duration<double>(b - a).count(); }

struct StreamRaces : RaceResolver {  // race pair -> prior matrix, shared by every chunk of the stream
  std::mutex mu;
  std::unordered_map<std::string, uint32_t> idx;
  std::vector<double> mats;  // [n][P*P]
  PriorSpec ps;
  uint32_t n = 0;
  uint32_t resolve(sv r1, sv r2) override {
    std::string key;
    key.reserve(r1.size() + r2.size() + 1);
    key.append(r1);
    key.push_back('\x01');
    key.append(r2);
    std::lock_guard<std::mutex> lk(mu);
    auto it = idx.find(key);
    if (it != idx.end()) return it->second;
    if (n >= 0xFFFEu) return 0xFFFFFFFFu;
    const size_t PP = ps.pops.size() * ps.pops.size();
    mats.resize((size_t)(n + 1) * PP);
    prior_matrix(ps, r1, r2, mats.data() + (size_t)n * PP);
    idx.emplace(std::move(key), n);
    return n++;
  }
};

// a chunk's text: a growable byte buffer that can be enlarged WITHOUT touching the new bytes (std::string::resize would
// zero a megabyte per chunk just to have it overwritten)
struct TextBuf {
  char *p = nullptr;      // the bytes: own_p, or memory of the caller's (a view)
  size_t n = 0, cap = 0;  // cap: of own_p
  char *own_p = nullptr;  // the buffer's own allocation: kept while the buffer is a view, used again afterwards
  // A VIEW (grim_stream_write_borrowed): the chunk's lines lie in a buffer the caller lends the stream until it is finished,
  // and nobody copies them before the tokenizer threads read them.  Never written, never freed; anything that wants to add
  // bytes takes a copy first (own()).
  bool view = false;
  TextBuf() {}
  TextBuf(const TextBuf &) = delete;
  TextBuf &operator=(const TextBuf &) = delete;
  ~TextBuf() { free(own_p); }
  const char *data() const { return p; }
  char *data() { return p; }
  size_t size() const { return n; }
  bool empty() const { return n == 0; }
  char back() const { return p[n - 1]; }
  void clear() {
    p = own_p;
    n = 0;
    view = false;
  }
  void set_view(const char *q, size_t k) {
    p = const_cast<char *>(q);
    n = k;
    view = true;
  }
  void grow(size_t want) {
    if (want <= cap) return;
    size_t c = cap ? cap : (1u << 16);
    while (c < want) c *= 2;
    char *q = (char *)realloc(own_p, c);
    if (!q) throw std::bad_alloc();
    own_p = q;
    cap = c;
  }
  void own() {  // a view becomes a copy
    if (!view) return;
    grow(n);
    if (n) memcpy(own_p, p, n);
    p = own_p;
    view = false;
  }
  void reserve(size_t want) {
    own();
    grow(want);
    p = own_p;
  }
  void resize(size_t want) {  // new bytes are NOT initialised
    reserve(want);
    n = want;
  }
  void append(const char *src, size_t k) {
    reserve(n + k);
    memcpy(p + n, src, k);
    n += k;
  }
};

enum { CH_FREE = 0, CH_FILLING, CH_TOKENIZING, CH_TOKENIZED, CH_RUN_DONE, CH_DEVICE_DONE, CH_FORMATTED, CH_COMMITTED, CH_WRITTEN };

struct Chunk {
  bool fetch_pending = false;  // the kernels are done, the results are still on the device
  bool in_flight = false;      // stage 1 of the whole chunk is enqueued: the copy thread waits for it
  std::vector<uint32_t> og_sorted;  // general-kernel subjects, heaviest first
  Clock::time_point t_dev0;
  Clock::time_point tl[10];  // GRIM_DEBUG_STREAM: when the chunk passed each stage (tl_note)
  uint32_t segment = 0;        // input segment the chunk belongs to (grim_stream_segment)
  int slot_no = 0;
  grim_batch *batch = nullptr;
  uint64_t index = 0, first_line = 0;
  TextBuf text;
  uint32_t n_lines = 0;
  std::vector<uint64_t> mark_off;  // byte offset of line k * granule
  // ranges
  uint32_t n_ranges = 0;
  std::vector<uint64_t> cut;               // [R + 1] byte boundaries
  std::vector<uint32_t> range_first_line;  // [R + 1]
  std::vector<uint64_t> slab_off, slab_cap;
  std::vector<TokRange> tr;
  std::vector<std::unique_ptr<FmtRange>> fr;
  std::vector<std::array<uint64_t, 7>> file_off;
  std::atomic<int> pending{0};
  int state = CH_FREE;
  // master class lists of the chunk (subject numbers = line numbers)
  std::vector<uint32_t> os, om, og;
  std::vector<SmallRec> small;
  std::vector<uint8_t> kinds;       // records mode
  std::vector<grim_row> extra_rows;  // rows of a chunk that had to be run in parts
  const grim_row *rows = nullptr;
  double device_s = 0;
  uint32_t n_dev_subjects = 0;
  uint32_t n_devtok = 0;  // lines left to the device tokenizer
  bool records_out = false, records_released = false;
};

// per-chunk stage times: slot taken, filled (dispatch), tokenised, device thread starts, launched, kernels seen done, export
// issued, results on the host, handed to the consumer, released -- summed over the chunks between consecutive stamps
static std::atomic<uint64_t> g_tl_ns[10];
static std::atomic<uint64_t> g_tl_n;
static inline void tl_note(Chunk *c, int k) { c->tl[k] = Clock::now(); }
static void tl_close(Chunk *c) {
  for (int k = 1; k < 10; ++k) g_tl_ns[k] += (uint64_t)(secs(c->tl[k - 1], c->tl[k]) * 1e9);
  ++g_tl_n;
}

struct Task {
  int type;  // 0 tokenize, 1 format, 2 write
  Chunk *c;
  uint32_t r;
};

}  // namespace

struct grim_stream {
  grim_ctx *ctx = nullptr;
  const grim_graph *graph = nullptr;
  grim_params prm;
  grim_stream_opts opt;
  DictSnap snap;
  StreamRaces races;
  MaskTable masks;
  bool have_masks = false;
  ClassRule rule;
  FmtParams fp;
  uint32_t n_pops = 0;
  uint32_t chunk_lines = 0, granule = 1024, n_threads = 1, depth = 0;
  uint64_t rows_per_chunk = 0;
  grim_devdict *devdict = nullptr;  // device tokenizer: the dictionary on the device (null: every line is tokenised on the host)
  bool dev_tok = false;
  std::atomic<uint64_t> rows_hint{0};  // row-pool size a chunk of this stream grew to (0: the opening size did)
  uint64_t rows_max = 0;               // ... and how far it may grow (0: the caller fixed the size)
  std::atomic<uint64_t> pool_hint{0};  // pair-pool records a chunk of this stream needed: every later load starts there
  // input segments (grim_stream_segment): cumulative text bytes at the end of each, known once its chunks are committed
  uint32_t cur_segment = 0;
  std::vector<std::array<uint64_t, 7>> seg_end;
  std::vector<uint8_t> seg_set;
  // placed output (opts.placed, grim/shard.py): the files are shared with the other ranks of a job, so where a segment's
  // piece of every file begins is only known once the pieces before it -- other ranks' -- have their sizes.  A committed chunk
  // hands its formatted buffers to its segment and is done; the buffers are written when the caller places the segment.
  struct SegBuf {
    int k;
    char *p;
    size_t n;
    uint64_t rel;  // offset inside the segment's piece of file k
  };
  struct PlacedWrite {
    int k;
    char *p;
    size_t n;
    uint64_t off;
    std::atomic<int> *left;
  };
  bool placed = false;
  std::vector<std::vector<SegBuf>> seg_bufs;
  std::vector<std::array<uint64_t, 7>> seg_begin;  // cumulative text bytes at the segment's first commit
  std::vector<uint8_t> seg_begun;
  std::vector<uint64_t> seg_first_chunk;  // index of the segment's first chunk
  std::deque<PlacedWrite> q_place;
  std::condition_variable cv_place;
  std::vector<std::string> stale;  // earlier runs' output files, moved aside at open
  std::thread unlinker;

  std::vector<std::unique_ptr<Chunk>> chunks;  // the `depth` slots
  Chunk *filling = nullptr;
  uint64_t next_index = 0, next_line = 0;
  bool carry_cr = false;

  std::mutex mu;  // guards task queues, chunk states, commit state
  std::condition_variable cv_work, cv_dev, cv_slot, cv_rec, cv_done;
  std::deque<Task> q_hi, q_lo;
  bool stop = false, input_closed = false;
  std::atomic<bool> failed{false};
  std::string err;
  std::vector<std::thread> workers;
  // the reader's copy helpers: the caller's bytes are cold memory and one thread copies them at 8-10 GB/s -- slower than
  // every other stage once GL parsing is on the device -- so a chunk's worth of input is copied by several threads at once
  struct CopyJob {
    const char *src = nullptr;
    char *dst = nullptr;
    size_t n = 0;
    std::vector<uint32_t> nl;  // out: offsets (inside the piece) of the piece's line ends
    uint32_t n_nl = 0;
  };
  std::vector<std::thread> copiers;
  std::vector<CopyJob> copy_jobs;  // the pieces of the block in hand (COPY_PIECES at most), claimed one by one through copy_next by the
                                   // reader and its helpers alike: a helper that is late, or that the host took away for a while,
                                   // copies fewer pieces instead of holding the chunk up with its fixed share
  // A claim is tied to its block: the ticket is (block number << 32 | next piece), the piece count is stored with the same
  // block number.  A helper whose last fetch_add was an over-claim on block g and that is descheduled before it looks at the
  // piece count can no longer take its stale index for a piece of block g + 1 (the count's block number differs: the claim
  // is dropped) -- and a claim (g, k) with k < count(g) keeps copy_left above zero, so block g + 1 cannot be posted under it.
  std::atomic<uint64_t> copy_ticket{0}, copy_np_gen{0};
  std::mutex copy_mu;
  std::condition_variable cv_copyjob;
  std::atomic<uint64_t> copy_gen{0};  // bumped (release) when the slots hold new jobs
  std::atomic<int> copy_left{0};
  bool copy_stop = false;
  uint32_t avg_line = 128;         // bytes per line of the last chunk: how much to copy before counting
  std::thread dev_thread, copy_thread, fetch_thread;
  std::deque<Chunk *> copy_q, fetch_q;
  std::condition_variable cv_copy, cv_fetch;
  uint64_t next_device = 0, next_commit = 0, next_record = 0, n_done = 0;

  int fd[6] = {-1, -1, -1, -1, -1, -1};
  uint64_t file_pos[7] = {0, 0, 0, 0, 0, 0, 0};
  std::vector<std::pair<char *, size_t>> mem[7];  // in-memory sinks: buffers in commit order
  std::string joined[7];
  bool joined_ok[7] = {false, false, false, false, false, false, false};

  struct Unsup {
    uint64_t line;
    uint32_t reason;
    std::string id;
  };
  std::vector<Unsup> unsupported;

  Clock::time_point t_open;
  grim_stream_stats st;
  std::atomic<uint64_t> tok_ns{0}, fmt_ns{0}, wr_ns{0};
  std::atomic<uint64_t> handed_back{0};  // lines the device tokenizer gave back to the host's

  void fail(const std::string &m) {  // mu held
    if (!failed) {
      failed = true;
      err = m;
    }
    cv_work.notify_all();
    cv_dev.notify_all();
    cv_copy.notify_all();
    cv_fetch.notify_all();
    cv_slot.notify_all();
    cv_rec.notify_all();
    cv_done.notify_all();
  }
  Chunk *by_index(uint64_t idx) {
    for (auto &c : chunks)
      if (c->state != CH_FREE && c->index == idx) return c.get();
    return nullptr;
  }
};

static void release_if_done(grim_stream *s, Chunk *c) {  // mu held
  if (c->state != CH_WRITTEN) return;
  if (s->opt.want_records && !c->records_released) return;
  c->state = CH_FREE;
  ++s->n_done;
  s->cv_slot.notify_all();
  s->cv_done.notify_all();
}

// ---- stage: tokenize one range ---------------------------------------------------------------------------------------
static void assemble(grim_stream *s, Chunk *c) {
  c->os.clear();
  c->om.clear();
  c->og.clear();
  c->small.clear();
  c->n_dev_subjects = 0;
  c->n_devtok = 0;
  bool race_overflow = false;
  for (uint32_t r = 0; r < c->n_ranges; ++r) {
    TokRange &T = c->tr[r];
    c->os.insert(c->os.end(), T.os.begin(), T.os.end());
    c->small.insert(c->small.end(), T.small.begin(), T.small.end());
    c->om.insert(c->om.end(), T.om.begin(), T.om.end());
    c->og.insert(c->og.end(), T.og.begin(), T.og.end());
    c->n_dev_subjects += T.n_subj;
    c->n_devtok += T.n_devtok;
    race_overflow = race_overflow || T.race_overflow;
  }
  if (s->opt.want_records) {
    c->kinds.clear();
    for (uint32_t r = 0; r < c->n_ranges; ++r) c->kinds.insert(c->kinds.end(), c->tr[r].kind.begin(), c->tr[r].kind.end());
  }
  std::lock_guard<std::mutex> lk(s->mu);
  if (race_overflow) s->fail("more than 65534 distinct race pairs in one run");
  c->state = CH_TOKENIZED;
  tl_note(c, 2);
  s->cv_dev.notify_all();
}

static void run_tokenize(grim_stream *s, Chunk *c, uint32_t r) {
  const auto t0 = Clock::now();
  const EngineHost *H = engine_batch_host(c->batch);
  TokRange &T = c->tr[r];
  T.clear();
  T.dense = false;
  T.first_subject = c->range_first_line[r];
  T.subj_dst = H->subj + c->range_first_line[r];
  T.tok_dst = H->tok + c->slab_off[r];
  T.tok_cap = c->slab_cap[r];
  T.tok_base = c->slab_off[r];
  TokParams tp{&s->snap, s->prm.planb != 0, &s->races, s->have_masks ? &s->masks : nullptr, &s->rule};
  if (s->dev_tok && H->lines && H->text) {
    tp.line_dst = H->lines + c->range_first_line[r];
    memcpy(H->text + c->cut[r], c->text.data() + c->cut[r], c->cut[r + 1] - c->cut[r]);  // the device reads the GL fields from here
  }
  tokenize_range(tp, c->text.data(), c->cut[r], c->cut[r + 1], T);
  const uint32_t expect = c->range_first_line[r + 1] - c->range_first_line[r];
  if (T.kind.size() != expect) {
    std::lock_guard<std::mutex> lk(s->mu);
    s->fail("internal: a range's line count differs from the reader's");
  }
  s->tok_ns += (uint64_t)(secs(t0, Clock::now()) * 1e9);
  if (c->pending.fetch_sub(1) == 1) assemble(s, c);
}

// ---- stage: device ------------------------------------------------------------------------------------------------------
// The device thread only STAGES and LAUNCHES: class lists into the pinned arena, one H2D, the kernels of stage 1, and on to
// the next chunk -- it never waits for the GPU.  The copy thread waits for a chunk's kernels (engine_batch_wait: an event
// behind its last kernel, not the stream), runs the second stage when subjects or accepted pairs are waiting for it,
// brings the results over on the copy stream and hands the chunk to the formatter.  A chunk whose results overflow a pool
// is run again by the copy thread, synchronously: the pair pool grown to what the run asked for, else in halves (a single
// subject always fits: the row pool is never smaller than one subject's worst case).
static std::atomic<uint64_t> g_dbg_ns[12];  // [4] reader: scan + copy + dispatch, [5] reader: waiting for a free chunk slot, [6] next_records: waiting  // GRIM_DEBUG_STREAM: staging / load + launch (device thread), wait + stage 2 / fetch (copy thread)

struct PartStats {  // what a part's run adds to the stream's statistics (applied under the lock by the caller)
  double kernel_ms[7] = {0, 0, 0, 0, 0, 0, 0};
  uint64_t counters[4] = {0, 0, 0, 0};
  uint64_t reruns = 0;
};

// lays the subjects of lines [lo, hi) out in the chunk's pinned arena and starts the H2D copy
static int stage_part(grim_stream *s, Chunk *c, uint32_t lo, uint32_t hi, size_t (&rng)[4], uint32_t &ng) {
  const auto tp0 = Clock::now();
  grim_batch *b = c->batch;
  const EngineHost *H = engine_batch_host(b);
  auto lb = [](const std::vector<uint32_t> &v, uint32_t x) { return (size_t)(std::lower_bound(v.begin(), v.end(), x) - v.begin()); };
  const size_t s0 = lb(c->os, lo), s1 = lb(c->os, hi), m0 = lb(c->om, lo), m1 = lb(c->om, hi);
  rng[0] = s0; rng[1] = s1; rng[2] = m0; rng[3] = m1;
  if (s1 > s0) {
    memcpy(H->small, c->small.data() + s0, sizeof(SmallRec) * (s1 - s0));
    memcpy(H->order_s, c->os.data() + s0, 4 * (s1 - s0));
  }
  if (m1 > m0) memcpy(H->order_m, c->om.data() + m0, 4 * (m1 - m0));
  ng = 0;
  for (uint32_t si : c->og_sorted)
    if (si >= lo && si < hi) H->order_g[ng++] = si;
  uint64_t tok_used = 0;
  for (uint32_t r = 0; r < c->n_ranges; ++r)
    if (c->tr[r].n_tok && c->range_first_line[r] < hi && c->range_first_line[r + 1] > lo) tok_used = c->slab_off[r] + c->tr[r].n_tok;
  const auto tp1 = Clock::now();
  engine_batch_hint_pool(b, s->pool_hint.load());  // what an earlier chunk of this stream had to grow its pair pool to
  engine_batch_hint_rows(b, s->rows_hint.load());  // ... and its row pool
  int lrc;
  {
    // the race table only grows: a batch that holds fewer matrices than the table has gets all of them again
    std::lock_guard<std::mutex> lk(s->races.mu);
    EngineLoad ld{c->n_lines, (uint32_t)(s1 - s0), (uint32_t)(m1 - m0), ng, tok_used, s->races.n, s->races.mats.data(), 0, 0, 0};
    if (c->n_devtok) {  // the lines of [lo, hi) whose record says so are tokenised on the device
      ld.dev_lo = lo;
      ld.dev_hi = hi;
      ld.text_bytes = c->text.size();
    }
    static const double one = 1.0;
    if (!s->races.n) {  // no line had two parsable fields: nothing indexes a matrix, the engine still wants one
      ld.n_priors = 0;
      ld.priors = &one;
    }
    lrc = engine_batch_load(b, &ld);
  }
  g_dbg_ns[0] += (uint64_t)(secs(tp0, tp1) * 1e9);
  g_dbg_ns[1] += (uint64_t)(secs(tp1, Clock::now()) * 1e9);
  return lrc;
}

static void part_stats(grim_stream *s, grim_batch *b, PartStats &ps) {
  if (!s->opt.timing) return;
  for (int k = 0; k < 7; ++k) ps.kernel_ms[k] += grim_batch_kernel_ms(b, k);
  uint64_t ctr[4];
  if (grim_batch_counters(b, ctr) == 0)
    for (int k = 0; k < 4; ++k) ps.counters[k] += ctr[k];
}

// synchronous run of the subjects of lines [lo, hi) (copy thread: the second try of a chunk, or a part of it)
static int device_part_sync(grim_stream *s, Chunk *c, uint32_t lo, uint32_t hi, bool whole, PartStats &ps);

// The device tokenizer handed lines of [lo, hi) back (an allele the graph does not know, 'g' / 'L' characters, loci out of
// order, ...: grim_tokdev.h).  The run's results are brought over as a PART of the chunk, the flagged lines go through the
// host's tokenizer after all (tokenize_lines) and -- those that are subjects -- through the device as a second small run,
// whose rows land behind the first run's.  Rare by construction; never a different answer, only a second pass.
static int fixup_part(grim_stream *s, Chunk *c, uint32_t lo, uint32_t hi, PartStats &ps) {
  grim_batch *b = c->batch;
  const uint32_t nrows1 = grim_batch_total_rows(b);
  const size_t base1 = c->extra_rows.size();
  c->extra_rows.resize(base1 + nrows1);
  if (engine_batch_fetch(b, lo, hi, nrows1 ? c->extra_rows.data() + base1 : nullptr) != 0) return -1;
  const EngineHost *H = engine_batch_host(b);
  grim_subject_result *res = H->res;
  std::vector<uint32_t> fos, fom, fog;
  std::vector<SmallRec> fsmall;
  std::vector<uint8_t> fixed(hi - lo, 0);
  TokParams tp{&s->snap, s->prm.planb != 0, &s->races, nullptr, &s->rule};
  uint64_t n_back = 0;
  for (uint32_t r = 0; r < c->n_ranges; ++r) {
    const uint32_t f0 = c->range_first_line[r], f1 = c->range_first_line[r + 1];
    if (f1 <= lo || f0 >= hi) continue;
    std::vector<uint32_t> list;
    for (uint32_t g = std::max(f0, lo); g < std::min(f1, hi); ++g)
      if (H->lines[g].gl_len && c->tr[r].kind[g - f0] == K_DEV && res[g].status == GRIM_ST_UNSUPPORTED && res[g].reason == 7) {
        list.push_back(g - f0);
        fixed[g - lo] = 1;
      }
    if (list.empty()) continue;
    n_back += list.size();
    tokenize_lines(tp, c->text.data(), c->tr[r], list, H->lines + f0, fos, fom, fog, fsmall);
    for (uint32_t j : list) {
      H->lines[f0 + j].gl_len = 0;  // no longer the device tokenizer's
      if (s->opt.want_records && f0 + j < c->kinds.size()) c->kinds[f0 + j] = c->tr[r].kind[j];
    }
  }
  s->handed_back += n_back;
  size_t base2 = c->extra_rows.size();
  if (!fos.empty() || !fom.empty() || !fog.empty()) {
    if (!fos.empty()) {
      memcpy(H->small, fsmall.data(), sizeof(SmallRec) * fsmall.size());
      memcpy(H->order_s, fos.data(), 4 * fos.size());
    }
    if (!fom.empty()) memcpy(H->order_m, fom.data(), 4 * fom.size());
    if (!fog.empty()) memcpy(H->order_g, fog.data(), 4 * fog.size());
    uint64_t tok_used = 0;
    for (uint32_t r = 0; r < c->n_ranges; ++r)
      if (c->tr[r].n_tok) tok_used = c->slab_off[r] + c->tr[r].n_tok;
    int rc;
    for (int attempt = 0;; ++attempt) {
      {
        std::lock_guard<std::mutex> lk(s->races.mu);
        EngineLoad ld{c->n_lines, (uint32_t)fos.size(), (uint32_t)fom.size(), (uint32_t)fog.size(), tok_used, s->races.n, s->races.mats.data(), 0, 0, 0};
        rc = engine_batch_load(b, &ld);
      }
      if (rc == 0) rc = grim_batch_run(b);  // (grows the pair pool and runs again by itself when a subject needs it)
      // the ROWS ran out: what finish_part does for a first run -- twice the pool, up to every line's worst case, and again
      // (the second run's inputs are still in the pinned arena)
      if (rc != -2 || attempt >= 8 || !s->rows_max || !engine_batch_grow_rows(b, s->rows_max)) break;
      ++ps.reruns;
      const uint64_t have = engine_batch_row_limit(b);
      uint64_t cur = s->rows_hint.load();
      while (have > cur && !s->rows_hint.compare_exchange_weak(cur, have)) {
      }
    }
    if (rc != 0) return -1;
    part_stats(s, b, ps);
    const uint32_t nrows2 = grim_batch_total_rows(b);
    c->extra_rows.resize(base2 + nrows2);
    // the headers again: the second run rewrote those of its subjects, the others still hold the first run's values
    if (engine_batch_fetch(b, lo, hi, nrows2 ? c->extra_rows.data() + base2 : nullptr) != 0) return -1;
    res = engine_batch_host(b)->res;
  }
  for (uint32_t r = 0; r < c->n_ranges; ++r) {
    const uint32_t f0 = c->range_first_line[r], f1 = c->range_first_line[r + 1];
    for (uint32_t g = std::max(f0, lo); g < std::min(f1, hi); ++g) {
      if (c->tr[r].kind[g - f0] != K_DEV) continue;
      const uint32_t off = (uint32_t)(fixed[g - lo] ? base2 : base1);
      if (off)
        for (int t = 0; t < GRIM_T_COUNT; ++t) res[g].row_off[t] += off;
    }
  }
  c->fetch_pending = false;
  return 0;
}

// after a run of [lo, hi) ended with `rc`: results stay on the device (whole chunk) or are fetched as a part; a pool that
// overflowed sends the range through device_part_sync again
static int finish_part(grim_stream *s, Chunk *c, uint32_t lo, uint32_t hi, bool whole, int rc, const size_t (&rng)[4], uint32_t ng, PartStats &ps) {
  grim_batch *b = c->batch;
  part_stats(s, b, ps);
  if (getenv("GRIM_DEBUG_STREAM") && atoi(getenv("GRIM_DEBUG_STREAM")) > 1)
    fprintf(stderr, "grim stream: chunk %llu lines [%u,%u) small %zu medium %zu general %u -> rc %d, rows %u (pool %llu)\n",
            (unsigned long long)c->index, lo, hi, rng[1] - rng[0], rng[3] - rng[2], ng, rc, grim_batch_total_rows(b), (unsigned long long)engine_batch_row_limit(b));
  if (rc == -2) {
    // subjects with tens of thousands of accepted pairs: give the table kernels' pair pool what the run asked for (up to
    // 256 M records, 21 GB with the arrays sized by it) and run the same subjects again, before halving the batch
    ++ps.reruns;
    if (engine_batch_grow_pool(b, 256ull << 20)) return device_part_sync(s, c, lo, hi, whole, ps);
    // the ROWS ran out (subjects that fill their tables: ~40 rows each where the pool opens with 32 per line): twice the pool,
    // for this slot now and for the others at their next load, up to every subject's worst case -- a stream of such
    // subjects would otherwise run every chunk three times (whole, then in halves)
    if (s->rows_max && engine_batch_grow_rows(b, s->rows_max)) {
      const uint64_t have = engine_batch_row_limit(b);
      uint64_t cur = s->rows_hint.load();
      while (have > cur && !s->rows_hint.compare_exchange_weak(cur, have)) {
      }
      return device_part_sync(s, c, lo, hi, whole, ps);
    }
    if (hi - lo <= 1) return -1;  // cannot happen: see grim_stream_open
    const uint32_t mid = lo + (hi - lo) / 2;
    const int r1 = device_part_sync(s, c, lo, mid, false, ps);
    if (r1 != 0) return r1;
    return device_part_sync(s, c, mid, hi, false, ps);
  }
  if (rc != 0) return -1;
  {  // this batch had to grow its pair pool: the other slots will meet the same kind of chunks
    const uint64_t want = engine_batch_pool_want(b);
    uint64_t cur = s->pool_hint.load();
    while (want > cur && !s->pool_hint.compare_exchange_weak(cur, want)) {
    }
  }
  if (engine_batch_irregular(b)) return fixup_part(s, c, lo, hi, ps);
  const uint32_t nrows = grim_batch_total_rows(b);
  if (whole) {
    c->fetch_pending = true;
  } else {
    // a part of a chunk: its rows are appended to the chunk's own array and the row offsets of its subjects re-based
    const size_t base = c->extra_rows.size();
    c->extra_rows.resize(base + nrows);
    if (engine_batch_fetch(b, lo, hi, nrows ? c->extra_rows.data() + base : nullptr) != 0) return -1;
    grim_subject_result *res = engine_batch_host(b)->res;
    if (base)  // every device subject of the lines [lo, hi): host- and device-tokenised alike
      for (uint32_t r = 0; r < c->n_ranges; ++r) {
        const uint32_t f0 = c->range_first_line[r], f1 = c->range_first_line[r + 1];
        for (uint32_t g = std::max(f0, lo); g < std::min(f1, hi); ++g)
          if (c->tr[r].kind[g - f0] == K_DEV)
            for (int t = 0; t < GRIM_T_COUNT; ++t) res[g].row_off[t] += (uint32_t)base;
      }
  }
  return 0;
}

static int device_part_sync(grim_stream *s, Chunk *c, uint32_t lo, uint32_t hi, bool whole, PartStats &ps) {
  size_t rng[4];
  uint32_t ng = 0;
  if (stage_part(s, c, lo, hi, rng, ng) != 0) return -1;
  int rc = engine_batch_enqueue(c->batch);
  if (rc == 0) rc = engine_batch_wait(c->batch);
  return finish_part(s, c, lo, hi, whole, rc, rng, ng, ps);
}

static void enqueue_format(grim_stream *s, Chunk *c);

// The device thread.  A chunk's input goes up as soon as the chunk is tokenised; its kernels are launched when the copy has
// arrived -- the thread watches it: the kernels could not start earlier anyway -- or, when another chunk turns up first, after
// that chunk's input has been sent up as well.  Either way the host has seen the upload finish before it launches, and the
// launch stream needs no wait for the upload stream's event (engine_batch_enqueue): that wait, a barrier packet on another
// queue's signal, costs the launch stream ~9 us per chunk (profiles/r4_notes.md).  Chunks reach the copy thread in input order.
static void device_loop(grim_stream *s) {
  static const bool eager = getenv("GRIM_EAGER_LAUNCH") && atoi(getenv("GRIM_EAGER_LAUNCH"));  // test switch: launch at once
  Chunk *loaded = nullptr;  // input sent up, kernels not launched yet
  auto launch = [&](Chunk *c, int rc) -> bool {  // kernels of a loaded chunk, then on to the copy thread
    const auto t0 = Clock::now();
    if (rc == 0 && c->n_dev_subjects) {
      rc = engine_batch_enqueue(c->batch);
      if (rc == 0) c->in_flight = true;
      else if (rc == -2) rc = 0;  // a pool too small before anything ran: the copy thread sorts it out (in_flight stays false)
    }
    const double busy = secs(t0, Clock::now());
    std::lock_guard<std::mutex> lk(s->mu);
    s->st.device_s += busy;
    s->st.subjects += c->n_dev_subjects;
    if (rc != 0) {
      const char *e = grim_last_error(s->ctx);
      s->fail(std::string("device stage failed: ") + (e ? e : ""));
      return false;
    }
    c->state = CH_RUN_DONE;
    tl_note(c, 4);
    s->copy_q.push_back(c);
    s->cv_copy.notify_all();
    return true;
  };
  for (;;) {
    Chunk *c = nullptr;
    {
      std::unique_lock<std::mutex> lk(s->mu);
      for (;;) {
        if (s->stop || s->failed) return;
        c = s->by_index(s->next_device);
        if (c && c->state == CH_TOKENIZED) break;
        c = nullptr;
        if (loaded) break;  // nothing else is ready: see to the chunk that is loaded
        s->cv_dev.wait(lk);
      }
      if (c) ++s->next_device;
    }
    if (!c && loaded && !eager) {
      // watch the copy (tens of microseconds) and the queue; whichever comes first -- the copy done: launch; another chunk
      // tokenised: send that one's input up first
      const auto w0 = Clock::now();
      for (int spin = 0;; ++spin) {
        if (engine_batch_upload_done(loaded->batch)) break;
        if ((spin & 7) == 7) {
          if (secs(w0, Clock::now()) > 200e-6) break;  // (a copy that takes this long: let the launch stream wait for it)
          std::unique_lock<std::mutex> lk(s->mu, std::try_to_lock);
          if (lk.owns_lock()) {
            if (s->stop || s->failed) return;
            Chunk *n = s->by_index(s->next_device);
            if (n && n->state == CH_TOKENIZED) {
              c = n;
              ++s->next_device;
              break;
            }
          }
        }
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
      }
    }
    int rc = 0;
    if (c) {
      const auto t0 = Clock::now();
      c->t_dev0 = t0;
      c->tl[3] = t0;
      c->extra_rows.clear();
      c->rows = nullptr;
      c->fetch_pending = false;
      c->in_flight = false;
      if (c->n_dev_subjects) {
        // longest-processing-time-first for the general kernel; stable: equal-cost subjects stay in input order
        const std::vector<uint32_t> &og = c->og;
        const grim_subject *subj = engine_batch_host(c->batch)->subj;
        std::vector<double> cost(og.size());
        for (size_t k = 0; k < og.size(); ++k) cost[k] = grim_cost(subj[og[k]]);
        std::vector<uint32_t> perm(og.size());
        for (size_t k = 0; k < perm.size(); ++k) perm[k] = (uint32_t)k;
        std::stable_sort(perm.begin(), perm.end(), [&](uint32_t a, uint32_t b) { return cost[a] > cost[b]; });
        c->og_sorted.resize(og.size());
        for (size_t k = 0; k < perm.size(); ++k) c->og_sorted[k] = og[perm[k]];
        size_t rng[4];
        uint32_t ng = 0;
        rc = stage_part(s, c, 0, c->n_lines, rng, ng);  // (ends with engine_batch_load: the H2D copy on the upload stream)
      }
      const double busy = secs(t0, Clock::now());
      std::lock_guard<std::mutex> lk(s->mu);
      s->st.device_s += busy;
    }
    if (loaded) {
      Chunk *l = loaded;
      loaded = nullptr;
      if (!launch(l, 0)) return;
    }
    if (c) {
      if (c->n_dev_subjects && rc == 0) {
        loaded = c;
      } else if (!launch(c, rc)) {  // nothing for the device (or a failed load: reported there)
        return;
      }
    }
  }
}

// the copy thread: waits for a chunk's kernels, second stage, D2H over the context's copy stream
static void copy_loop(grim_stream *s) {
  for (;;) {
    Chunk *c = nullptr;
    {
      std::unique_lock<std::mutex> lk(s->mu);
      for (;;) {
        if (s->stop || s->failed) return;
        if (!s->copy_q.empty()) {
          c = s->copy_q.front();
          s->copy_q.pop_front();
          break;
        }
        s->cv_copy.wait(lk);
      }
    }
    const auto t0 = Clock::now();
    int rc = 0;
    PartStats ps;
    if (c->n_dev_subjects) {
      if (c->in_flight) {
        c->in_flight = false;
        const int wrc = engine_batch_wait(c->batch);
        tl_note(c, 5);
        auto lb = [](const std::vector<uint32_t> &v, uint32_t x) { return (size_t)(std::lower_bound(v.begin(), v.end(), x) - v.begin()); };
        const size_t rng[4] = {0, lb(c->os, c->n_lines), 0, lb(c->om, c->n_lines)};
        rc = finish_part(s, c, 0, c->n_lines, true, wrc, rng, (uint32_t)c->og_sorted.size(), ps);
      } else {
        rc = device_part_sync(s, c, 0, c->n_lines, true, ps);
      }
      if (rc == 0 && !c->extra_rows.empty()) c->rows = c->extra_rows.data();
    }
    const auto t1 = Clock::now();
    g_dbg_ns[2] += (uint64_t)(secs(t0, t1) * 1e9);
    // the results come over on the copy stream; the fetch thread waits for them, this one goes back to watching kernels
    bool issued = false, fetched_ok = true;
    if (rc == 0 && c->n_dev_subjects && c->fetch_pending && c->extra_rows.empty()) {
      fetched_ok = engine_batch_fetch_issue(c->batch) == 0;
      issued = fetched_ok;
    }
    c->device_s = secs(c->t_dev0, Clock::now());
    tl_note(c, 6);
    {
      std::lock_guard<std::mutex> lk(s->mu);
      for (int k = 0; k < 7; ++k) s->st.kernel_ms[k] += ps.kernel_ms[k];
      for (int k = 0; k < 4; ++k) s->st.counters[k] += ps.counters[k];
      s->st.reruns += ps.reruns;
      if (rc != 0 || !fetched_ok) {
        const char *e = grim_last_error(s->ctx);
        s->fail(std::string(rc != 0 ? "device stage failed: " : "device stage failed: copying the results to the host: ") + (e ? e : ""));
        return;
      }
      if (issued) {
        s->fetch_q.push_back(c);
        s->cv_fetch.notify_all();
      } else {
        c->state = CH_DEVICE_DONE;
      }
    }
    if (!issued) enqueue_format(s, c);
  }
}

// the fetch thread: waits for a chunk's D2H copy and hands the chunk to the formatter
static void fetch_loop(grim_stream *s) {
  for (;;) {
    Chunk *c = nullptr;
    {
      std::unique_lock<std::mutex> lk(s->mu);
      for (;;) {
        if (s->stop || s->failed) return;
        if (!s->fetch_q.empty()) {
          c = s->fetch_q.front();
          s->fetch_q.pop_front();
          break;
        }
        s->cv_fetch.wait(lk);
      }
    }
    const auto t0 = Clock::now();
    const bool ok = engine_batch_fetch_wait(c->batch) == 0;
    tl_note(c, 7);
    g_dbg_ns[3] += (uint64_t)(secs(t0, Clock::now()) * 1e9);
    {
      std::lock_guard<std::mutex> lk(s->mu);
      if (!ok) {
        const char *e = grim_last_error(s->ctx);
        s->fail(std::string("device stage failed: copying the results to the host: ") + (e ? e : ""));
        return;
      }
      c->rows = engine_batch_host(c->batch)->rows;
      c->state = CH_DEVICE_DONE;
    }
    enqueue_format(s, c);
  }
}

// ---- stage: format, commit, write ------------------------------------------------------------------------------------
static void commit_ready(grim_stream *s) {  // mu held: commits every chunk that is formatted, in input order
  for (;;) {
    Chunk *c = s->by_index(s->next_commit);
    if (!c || c->state != CH_FORMATTED) return;
    const grim_subject_result *res = engine_batch_host(c->batch)->res;
    bool any_file = false;
    if (c->segment < s->seg_begun.size() && !s->seg_begun[c->segment]) {
      s->seg_begun[c->segment] = 1;
      for (int k = 0; k < 7; ++k) s->seg_begin[c->segment][k] = s->file_pos[k];
    }
    for (uint32_t r = 0; r < c->n_ranges; ++r) {
      FmtRange &F = *c->fr[r];
      for (int k = 0; k < 7; ++k) {
        c->file_off[r][k] = s->file_pos[k];
        s->file_pos[k] += F.t[k].n;
        s->st.text_bytes[k] += F.t[k].n;
        if (k < 6 && s->fd[k] >= 0 && s->placed) {  // placed output: the buffer waits in its segment for the segment's base
          if (F.t[k].n) {
            s->seg_bufs[c->segment].push_back({k, F.t[k].p, F.t[k].n, c->file_off[r][k] - s->seg_begin[c->segment][k]});
            F.t[k].p = nullptr;
            F.t[k].n = F.t[k].cap = 0;
          }
        } else if (k < 6 && s->fd[k] >= 0) {
          any_file = any_file || F.t[k].n;
        } else if (F.t[k].n) {  // in-memory sink: the buffer changes hands
          s->mem[k].emplace_back(F.t[k].p, F.t[k].n);
          F.t[k].p = nullptr;
          F.t[k].n = F.t[k].cap = 0;
        }
      }
      for (uint32_t j : F.unsupported) {
        const TokRange &T = c->tr[r];
        uint32_t reason = T.kind[j] == K_UNSUPPORTED_GL ? 8 : 5;
        if (T.kind[j] == K_DEV) reason = res[c->range_first_line[r] + j].reason;
        s->unsupported.push_back({c->first_line + c->range_first_line[r] + j, reason,
                                  std::string(c->text.data() + T.line[j].off, T.line[j].id_len)});
      }
    }
    if (c->segment < s->seg_end.size()) {  // everything of this segment up to here is placed
      for (int k = 0; k < 7; ++k) s->seg_end[c->segment][k] = s->file_pos[k];
      s->seg_set[c->segment] = 1;
    }
    ++s->next_commit;
    c->state = CH_COMMITTED;
    s->cv_done.notify_all();  // (grim_stream_segment_wait)
    if (s->opt.want_records) s->cv_rec.notify_all();
    if (any_file) {
      c->pending.store((int)c->n_ranges);
      for (uint32_t r = 0; r < c->n_ranges; ++r) s->q_hi.push_back(Task{2, c, r});
      s->cv_work.notify_all();
    } else {
      c->state = CH_WRITTEN;
      release_if_done(s, c);
    }
  }
}

static void enqueue_format(grim_stream *s, Chunk *c) {
  std::lock_guard<std::mutex> lk(s->mu);
  if (!s->opt.want_text) {
    for (uint32_t r = 0; r < c->n_ranges; ++r) {
      c->fr[r]->unsupported.clear();
      for (int k = 0; k < 7; ++k) c->fr[r]->t[k].clear();
    }
    c->state = CH_FORMATTED;
    commit_ready(s);
    return;
  }
  c->pending.store((int)c->n_ranges);
  for (uint32_t r = 0; r < c->n_ranges; ++r) s->q_hi.push_back(Task{1, c, r});
  s->cv_work.notify_all();
}

static void run_format(grim_stream *s, Chunk *c, uint32_t r) {
  const auto t0 = Clock::now();
  FmtRange &F = *c->fr[r];
  for (int k = 0; k < 7; ++k) F.t[k].clear();
  F.unsupported.clear();
  FmtParams fp = s->fp;  // copies a few strings; per range, not per line
  fp.per_subject_s = c->n_dev_subjects ? c->device_s / c->n_dev_subjects : 0.0;
  format_range(fp, c->text.data(), c->tr[r], engine_batch_host(c->batch)->res, c->rows, c->first_line + c->range_first_line[r],
               nullptr, F);
  s->fmt_ns += (uint64_t)(secs(t0, Clock::now()) * 1e9);
  if (c->pending.fetch_sub(1) == 1) {
    std::lock_guard<std::mutex> lk(s->mu);
    c->state = CH_FORMATTED;
    commit_ready(s);
  }
}

static void run_write(grim_stream *s, Chunk *c, uint32_t r) {
  const auto t0 = Clock::now();
  FmtRange &F = *c->fr[r];
  bool ok = true;
  for (int k = 0; k < 6 && ok; ++k) {
    if (s->fd[k] < 0 || !F.t[k].n) continue;
    size_t done = 0;
    while (done < F.t[k].n) {
      const ssize_t w = pwrite(s->fd[k], F.t[k].p + done, F.t[k].n - done, (off_t)(c->file_off[r][k] + done));
      if (w < 0) {
        if (errno == EINTR) continue;
        ok = false;
        break;
      }
      done += (size_t)w;
    }
  }
  s->wr_ns += (uint64_t)(secs(t0, Clock::now()) * 1e9);
  if (!ok) {
    std::lock_guard<std::mutex> lk(s->mu);
    s->fail(std::string("writing an output file failed: ") + strerror(errno));
  }
  if (c->pending.fetch_sub(1) == 1) {
    std::lock_guard<std::mutex> lk(s->mu);
    c->state = CH_WRITTEN;
    release_if_done(s, c);
  }
}

// one buffer of a placed segment: written at its final position, then released
static void run_placed_write(grim_stream *s, const grim_stream::PlacedWrite &w) {
  const auto t0 = Clock::now();
  bool ok = true;
  size_t done = 0;
  while (done < w.n) {
    const ssize_t k = pwrite(s->fd[w.k], w.p + done, w.n - done, (off_t)(w.off + done));
    if (k < 0) {
      if (errno == EINTR) continue;
      ok = false;
      break;
    }
    done += (size_t)k;
  }
  const int e = errno;
  free(w.p);
  s->wr_ns += (uint64_t)(secs(t0, Clock::now()) * 1e9);
  std::lock_guard<std::mutex> lk(s->mu);
  if (!ok) s->fail(std::string("writing an output file failed: ") + strerror(e));
  if (w.left->fetch_sub(1) == 1) s->cv_place.notify_all();
}

static void worker_loop(grim_stream *s) {
  for (;;) {
    Task t;
    grim_stream::PlacedWrite pw{0, nullptr, 0, 0, nullptr};
    bool more = false, have_pw = false;
    {
      std::unique_lock<std::mutex> lk(s->mu);
      for (;;) {
        if (s->stop) return;
        if (!s->q_place.empty()) {
          pw = s->q_place.front();
          s->q_place.pop_front();
          have_pw = true;
          break;
        }
        if (!s->q_hi.empty()) {
          t = s->q_hi.front();
          s->q_hi.pop_front();
          break;
        }
        if (!s->q_lo.empty()) {
          t = s->q_lo.front();
          s->q_lo.pop_front();
          break;
        }
        s->cv_work.wait(lk);
      }
      more = !s->q_hi.empty() || !s->q_lo.empty() || !s->q_place.empty();
    }
    if (more) s->cv_work.notify_one();
    if (have_pw) {
      run_placed_write(s, pw);
      continue;
    }
    if (t.type == 0) run_tokenize(s, t.c, t.r);
    else if (t.type == 1) run_format(s, t.c, t.r);
    else run_write(s, t.c, t.r);
  }
}

// copies a piece and notes where its line ends are (offsets inside the piece), so that the reader does not have to look at
// the bytes again
#if defined(__x86_64__)
__attribute__((target("avx2"))) static uint32_t copy_and_mark_avx2(char *dst, const char *src, size_t n, uint32_t *nl) {
  const __m256i nlv = _mm256_set1_epi8('\n');
  uint32_t k = 0;
  size_t i = 0;
  for (; i + 32 <= n; i += 32) {
    const __m256i v = _mm256_loadu_si256((const __m256i *)(src + i));
    _mm256_storeu_si256((__m256i *)(dst + i), v);
    for (uint32_t m = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(v, nlv)); m; m &= m - 1) nl[k++] = (uint32_t)i + (uint32_t)__builtin_ctz(m);
  }
  for (; i < n; ++i) {
    dst[i] = src[i];
    if (src[i] == '\n') nl[k++] = (uint32_t)i;
  }
  return k;
}
#endif
#if defined(__x86_64__)
__attribute__((target("avx2"))) static uint32_t scan_and_mark_avx2(const char *src, size_t n, uint32_t *nl) {  // the same, nothing copied
  const __m256i nlv = _mm256_set1_epi8('\n');
  uint32_t k = 0;
  size_t i = 0;
  for (; i + 32 <= n; i += 32) {
    const __m256i v = _mm256_loadu_si256((const __m256i *)(src + i));
    for (uint32_t m = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(v, nlv)); m; m &= m - 1) nl[k++] = (uint32_t)i + (uint32_t)__builtin_ctz(m);
  }
  for (; i < n; ++i)
    if (src[i] == '\n') nl[k++] = (uint32_t)i;
  return k;
}
#endif
static uint32_t copy_and_mark(char *dst, const char *src, size_t n, uint32_t *nl) {
#if defined(__x86_64__)
  static const bool avx2 = __builtin_cpu_supports("avx2");
  if (avx2) return dst ? copy_and_mark_avx2(dst, src, n, nl) : scan_and_mark_avx2(src, n, nl);
#endif
  if (!dst) {
    uint32_t k = 0;
    for (size_t i = 0; i < n; ++i)
      if (src[i] == '\n') nl[k++] = (uint32_t)i;
    return k;
  }
  uint32_t k = 0;
  for (size_t i = 0; i < n; ++i) {
    dst[i] = src[i];
    if (src[i] == '\n') nl[k++] = (uint32_t)i;
  }
  return k;
}

// pieces of the posted block, until none is left (reader and helpers)
static constexpr size_t COPY_PIECES = 64;
static void copy_pieces(grim_stream *s) {
  for (;;) {
    // (a thread that comes back here from the block before may find the NEXT block posted already: the piece it takes then
    //  is one of that block's -- everything about a block is written before the ticket is reset with the block's number)
    const uint64_t t = s->copy_ticket.fetch_add(1, std::memory_order_acq_rel);
    const uint64_t ng = s->copy_np_gen.load(std::memory_order_acquire);
    const uint32_t k = (uint32_t)t;
    if ((ng >> 32) != (t >> 32) || k >= (uint32_t)ng) return;  // another block's count, or past the end of this one
    grim_stream::CopyJob &job = s->copy_jobs[k];
    tl_dbg[0] = k; tl_dbg[1] = (uint64_t)job.dst; tl_dbg[2] = (uint64_t)job.src; tl_dbg[3] = job.n; tl_dbg[4] = (uint64_t)job.nl.data();
    tl_dbg[5] = job.nl.size(); tl_dbg[6] = t; tl_dbg[7] = ng;
    job.n_nl = copy_and_mark(job.dst, job.src, job.n, job.nl.data());
    s->copy_left.fetch_sub(1, std::memory_order_release);
  }
}

// a copy helper of the reader (see grim_stream::CopyJob): waits until the reader posts a block, takes pieces of it
static void copier_loop(grim_stream *s, size_t) {
  uint64_t seen = 0;
  for (;;) {
    {
      // A stream in full flow posts a block every ~70 us, and a thread asleep on a condition variable takes tens of
      // microseconds to come back -- longer than its pieces take to copy.  So a helper keeps LOOKING for the next block for
      // a little while (250 us: three chunk periods) before it goes to sleep; an idle stream costs nothing.
      uint64_t gen = s->copy_gen.load(std::memory_order_acquire);
      if (gen == seen) {
        const auto t0 = Clock::now();
        for (int spin = 0; gen == seen; ++spin) {
          if ((spin & 63) == 63 && secs(t0, Clock::now()) > 250e-6) break;
#if defined(__x86_64__)
          __builtin_ia32_pause();
#endif
          gen = s->copy_gen.load(std::memory_order_acquire);
        }
      }
      if (gen == seen) {
        std::unique_lock<std::mutex> lk(s->copy_mu);
        s->cv_copyjob.wait(lk, [&] { return s->copy_stop || s->copy_gen.load(std::memory_order_acquire) != seen; });
        if (s->copy_stop) return;
        gen = s->copy_gen.load(std::memory_order_acquire);
      }
      seen = gen;
    }
    copy_pieces(s);  // (a helper that arrives after the block is done finds no piece: the ticket is past the end)
  }
}

// ---- the reader side: bytes -> chunks ----------------------------------------------------------------------------------
static Chunk *acquire_chunk(grim_stream *s) {
  std::unique_lock<std::mutex> lk(s->mu);
  for (;;) {
    if (s->failed) return nullptr;
    for (auto &c : s->chunks)
      if (c->state == CH_FREE) {
        c->state = CH_FILLING;
        tl_note(c.get(), 0);
        c->index = s->next_index++;
        c->first_line = s->next_line;
        c->text.clear();
        c->n_lines = 0;
        c->mark_off.clear();
        c->mark_off.push_back(0);  // line 0 starts at byte 0; line k * granule: noted when its predecessor ends
        c->records_out = c->records_released = false;
        return c.get();
      }
    s->cv_slot.wait(lk);
  }
}

static int dispatch(grim_stream *s, Chunk *c) {
  // ranges: whole granules, about one range per thread
  const uint32_t n = c->n_lines;
  uint32_t per = (n + s->n_threads - 1) / s->n_threads;
  per = ((per + s->granule - 1) / s->granule) * s->granule;
  if (per == 0) per = s->granule;
  const uint32_t R = (n + per - 1) / per;
  c->n_ranges = R;
  c->cut.assign(R + 1, c->text.size());
  c->range_first_line.assign(R + 1, n);
  c->slab_off.assign(R, 0);
  c->slab_cap.assign(R, 0);
  for (uint32_t r = 0; r < R; ++r) {
    c->range_first_line[r] = r * per;
    c->cut[r] = c->mark_off[(size_t)r * per / s->granule];
  }
  uint64_t tok_total = 0;
  for (uint32_t r = 0; r < R; ++r) {
    c->slab_off[r] = tok_total;
    c->slab_cap[r] = (c->cut[r + 1] - c->cut[r]) / 2 + 16;
    tok_total += c->slab_cap[r];
  }
  if (c->tr.size() < R) c->tr.resize(R);
  while (c->fr.size() < R) c->fr.emplace_back(new FmtRange());
  if (c->file_off.size() < R) c->file_off.resize(R);
  EnginePlan pl{n, tok_total, s->dev_tok ? (uint64_t)c->text.size() : 0ull};
  if (engine_batch_plan(c->batch, &pl) != 0) {
    std::lock_guard<std::mutex> lk(s->mu);
    s->fail(std::string("laying out a chunk's buffers failed: ") + grim_last_error(s->ctx));
    return -1;
  }
  s->next_line += n;
  std::lock_guard<std::mutex> lk(s->mu);
  c->segment = s->cur_segment;
  s->st.lines += n;
  ++s->st.chunks;
  c->state = CH_TOKENIZING;
  tl_note(c, 1);
  c->pending.store((int)R);
  for (uint32_t r = 0; r < R; ++r) s->q_lo.push_back(Task{0, c, r});
  s->cv_work.notify_one();  // (a worker that takes a task wakes the next one while tasks are left: waking all thirty from
                            // here cost the reader 50 us per chunk)
  return 0;
}

// The reader's line counter.  It only has to know where the chunk ends and where every `granule`-th line starts, so it counts
// line ends 32 bytes at a time and looks at single bytes only in the few blocks where one of those events falls (a memchr
// per line was 80 us of the reader's 125 us per 10 000-line chunk -- the stream's bottleneck once GL parsing had moved to the
// device).  Returns the lines found (at most `want`); *used = bytes up to and including the last line end found, or len
// when fewer than `want` were found.  have: lines the chunk holds already; base: offset of p[0] inside the chunk's text.
struct LineScan {
  uint32_t granule, chunk_lines;
  std::vector<uint64_t> *marks;
};
static inline bool scan_event(const LineScan &L, uint32_t ln, uint64_t off_after) {  // line end number ln (1-based in the chunk)
  if (ln % L.granule == 0 && ln < L.chunk_lines) L.marks->push_back(off_after);
  return false;
}
static uint32_t scan_lines_scalar(const LineScan &L, const char *p, uint64_t len, uint32_t want, uint32_t have, uint64_t base, uint64_t *used) {
  uint64_t b = 0;
  uint32_t got = 0;
  while (b < len && got < want) {
    const char *nl = (const char *)memchr(p + b, '\n', len - b);
    if (!nl) {
      b = len;
      break;
    }
    b = (uint64_t)(nl - p) + 1;
    ++got;
    scan_event(L, have + got, base + b);
  }
  *used = b;
  return got;
}
#if defined(__x86_64__)
__attribute__((target("avx2,popcnt"))) static uint32_t scan_lines_avx2(const LineScan &L, const char *p, uint64_t len, uint32_t want, uint32_t have,
                                                                        uint64_t base, uint64_t *used) {
  const __m256i nlv = _mm256_set1_epi8('\n');
  uint64_t i = 0;
  uint32_t got = 0;
  while (i + 32 <= len && got < want) {
    const uint32_t m = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_loadu_si256((const __m256i *)(p + i)), nlv));
    const uint32_t c = (uint32_t)_mm_popcnt_u32(m);
    const uint32_t ln0 = have + got;                                   // line ends before this block
    const uint32_t to_mark = L.granule - ln0 % L.granule;              // line ends until the next multiple of the granule
    if (c < to_mark && got + c < want) {                               // no event in this block
      got += c;
      i += 32;
      continue;
    }
    for (uint32_t mm = m; mm; mm &= mm - 1) {
      const uint32_t k = (uint32_t)__builtin_ctz(mm);
      ++got;
      scan_event(L, have + got, base + i + k + 1);
      if (got == want) {
        *used = i + k + 1;
        return got;
      }
    }
    i += 32;
  }
  if (got < want && i < len) {
    uint64_t u2 = 0;
    got += scan_lines_scalar(L, p + i, len - i, want - got, have + got, base + i, &u2);
    *used = i + u2;
    return got;
  }
  *used = got < want ? len : i;
  return got;
}
#endif
static uint32_t scan_lines(const LineScan &L, const char *p, uint64_t len, uint32_t want, uint32_t have, uint64_t base, uint64_t *used) {
#if defined(__x86_64__)
  static const bool avx2 = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("popcnt");
  if (avx2) return scan_lines_avx2(L, p, len, want, have, base, used);
#endif
  return scan_lines_scalar(L, p, len, want, have, base, used);
}

// Byte offsets of every chunk_lines-th line start of a file (and the file's size as the last entry): what a rank of
// grim/shard.py needs to pull "chunk c" of the input as a byte range.  Lines end as grim_stream_write_text ends them
// (universal newlines: "\n", "\r\n", a lone "\r"); a last line without its line end counts.  Returns the number of
// entries (>= 1), or -1 (errno says why); *out is malloc'd, the caller frees it with grim_free.
extern "C" int64_t grim_chunk_offsets(const char *path, uint32_t chunk_lines, uint64_t **out) {
  if (!path || !out || chunk_lines == 0) return -1;
  *out = nullptr;
  const int fd = open(path, O_RDONLY);
  if (fd < 0) return -1;
  struct stat sb;
  if (fstat(fd, &sb) != 0) {
    close(fd);
    return -1;
  }
  const uint64_t size = (uint64_t)sb.st_size;
  std::vector<uint64_t> offs;
  offs.push_back(0);
  if (size) {
    std::vector<char> buf(16u << 20);
    uint64_t pos = 0;
    uint32_t have = 0;  // line ends so far, modulo chunk_lines
    bool pending_cr = false;  // the last byte of the previous block was a '\r' that may be the first half of "\r\n"
    const LineScan L{chunk_lines, 0xFFFFFFFFu, &offs};
    for (;;) {
      const ssize_t n = read(fd, buf.data(), buf.size());
      if (n < 0) {
        if (errno == EINTR) continue;
        close(fd);
        return -1;
      }
      if (n == 0) break;
      char *p = buf.data();
      const uint64_t m = (uint64_t)n;
      if (!pending_cr && !memchr(p, '\r', m)) {  // the usual file: '\n' only -- the vector scan
        uint64_t done = 0;
        while (done < m) {
          uint64_t used = 0;
          const uint32_t got = scan_lines(L, p + done, m - done, 0x40000000u, have, pos + done, &used);
          have = (have + got) % chunk_lines;
          done += used;
          if (got < 0x40000000u) break;
        }
      } else {
        for (uint64_t i = 0; i < m; ++i) {
          const char ch = p[i];
          bool end = false;
          if (pending_cr) {  // the '\r' before this byte: a line end of its own unless this is its '\n'
            pending_cr = false;
            if (ch == '\n') {
              end = true;  // "\r\n": one line end, after the '\n'
            } else {
              have = (have + 1) % chunk_lines;
              if (have == 0) offs.push_back(pos + i);
            }
          }
          if (!end) {
            if (ch == '\n') end = true;
            else if (ch == '\r') pending_cr = true;
          }
          if (end) {
            have = (have + 1) % chunk_lines;
            if (have == 0) offs.push_back(pos + i + 1);
          }
        }
      }
      pos += m;
    }
    if (pending_cr) {
      have = (have + 1) % chunk_lines;
      if (have == 0) offs.push_back(pos);
    }
    if (offs.back() >= size && offs.size() > 1) offs.pop_back();
    offs.push_back(size);
  }
  close(fd);
  uint64_t *r = (uint64_t *)malloc(sizeof(uint64_t) * offs.size());
  if (!r) return -1;
  memcpy(r, offs.data(), sizeof(uint64_t) * offs.size());
  *out = r;
  return (int64_t)offs.size();
}
extern "C" void grim_free(void *p) { free(p); }

static int stream_write(grim_stream *s, const char *text, uint64_t len, bool borrowed);
extern "C" int grim_stream_write(grim_stream *s, const char *text, uint64_t len) { return stream_write(s, text, len, false); }
// The caller LENDS the bytes: they stay where they are, valid and unchanged, until grim_stream_finish has returned (or the
// stream is freed).  A chunk that begins inside such a buffer is a view of it: the reader and its helpers only find the line
// ends (a read of the cold bytes instead of a read and a write), the tokenizer threads read the lines where they lie.
extern "C" int grim_stream_write_borrowed(grim_stream *s, const char *text, uint64_t len) {
  static const bool off = getenv("GRIM_NO_BORROW") && atoi(getenv("GRIM_NO_BORROW"));  // test switch: copy after all
  return stream_write(s, text, len, !off);
}
static int stream_write(grim_stream *s, const char *text, uint64_t len, bool borrowed) {
  if (!s || s->input_closed) return -1;
  uint64_t a = 0;
  while (a < len) {
    const auto tr0 = Clock::now();
    if (!s->filling) {
      s->filling = acquire_chunk(s);
      if (!s->filling) return -1;
    }
    const auto tr1 = Clock::now();
    g_dbg_ns[5] += (uint64_t)(secs(tr0, tr1) * 1e9);
    Chunk *c = s->filling;
    // take whole lines until the chunk is full; a partial last line stays open (the next call continues it)
    // Copy first, count the lines in the COPY: the caller's bytes are cold (a file's pages, a buffer just filled), the copy
    // is in cache.  A big input is copied a chunk's worth at a time -- as many bytes as the chunk will probably need, going
    // by the last chunk's line length -- by this thread and its helpers together; what the copy holds beyond the chunk's last
    // line is dropped again and goes to the next chunk, what it lacks comes with the next turn of the loop.
    const uint32_t want = s->chunk_lines - c->n_lines;
    const uint64_t base = c->text.size();
    const uint64_t remaining = len - a;
    uint64_t blk;
    uint64_t used = 0;
    uint32_t got = 0;
    if (remaining >= (256u << 10) && !s->copiers.empty()) {
      blk = std::min<uint64_t>(remaining, (uint64_t)want * s->avg_line + 4096);
      const bool as_view = borrowed && base == 0 && c->n_lines == 0;  // the chunk begins here: its lines stay in the caller's buffer
      if (!as_view) c->text.resize(base + blk);
      // pieces of ~48 KB (whole cache lines), at least one per thread
      const size_t nh = s->copiers.size();
      size_t parts = std::min<size_t>(COPY_PIECES, std::max<size_t>(nh + 1, (size_t)(blk / (48u << 10))));
      // (ceiling first: with the floor, a quotient that is already a multiple of 64 left a remainder for a 65th piece -- one
      //  more than copy_jobs holds)
      const size_t per = (((blk + parts - 1) / parts) + 63) & ~(size_t)63;
      parts = (size_t)((blk + per - 1) / per);
      if (parts > COPY_PIECES) parts = COPY_PIECES;  // (cannot happen: per >= blk / parts)
      {
        std::lock_guard<std::mutex> lk(s->copy_mu);
        for (size_t k = 0; k < parts; ++k) {
          const size_t o = std::min<size_t>(blk, k * per), e = std::min<size_t>(blk, (k + 1) * per);
          grim_stream::CopyJob &j = s->copy_jobs[k];
          j.src = text + a + o;
          j.dst = as_view ? nullptr : c->text.data() + base + o;
          j.n = e - o;
          j.n_nl = 0;
          if (j.nl.size() < j.n + 64) j.nl.resize(j.n + 64);
        }
        const uint64_t blk_no = (s->copy_gen.load(std::memory_order_relaxed) + 1) & 0xFFFFFFFFull;  // only this thread writes copy_gen
        s->copy_np_gen.store(blk_no << 32 | (uint64_t)parts, std::memory_order_release);
        s->copy_left.store((int)parts);
        s->copy_ticket.store(blk_no << 32, std::memory_order_release);  // from here on pieces can be taken (no piece of the block
                                                                        // before is in flight: copy_left was 0)
        s->copy_gen.fetch_add(1, std::memory_order_release);  // ... and sleeping / looking helpers learn of the block
      }
      s->cv_copyjob.notify_all();
      copy_pieces(s);
      while (s->copy_left.load(std::memory_order_acquire) > 0) std::this_thread::yield();
      const auto tr2 = Clock::now();
      g_dbg_ns[8] += (uint64_t)(secs(tr1, tr2) * 1e9);
      // the pieces in order: the chunk ends behind its `want`-th line end; every granule-th line start is a mark
      used = blk;
      bool full = false;
      for (size_t q = 0; q < parts && !full; ++q) {
        const grim_stream::CopyJob &j = s->copy_jobs[q];
        const uint64_t po = std::min<size_t>(blk, q * per);
        const uint32_t take = std::min<uint32_t>(j.n_nl, want - got);
        // marks: line ends number `ln` (1-based in the chunk) with ln % granule == 0
        uint32_t ln = c->n_lines + got;
        uint32_t first = s->granule - ln % s->granule;  // the first of this piece's line ends that is a multiple
        for (uint32_t i = first; i <= take; i += s->granule)
          if (ln + i < s->chunk_lines) c->mark_off.push_back(base + po + j.nl[i - 1] + 1);
        got += take;
        if (got == want) {
          used = po + j.nl[take - 1] + 1;
          full = true;
        }
      }
      if (as_view) c->text.set_view(text + a, used);
      else c->text.resize(base + used);
      g_dbg_ns[9] += (uint64_t)(secs(tr2, Clock::now()) * 1e9);
    } else {
      blk = std::min<uint64_t>(remaining, 64u << 10);
      c->text.append(text + a, blk);
      const LineScan L{s->granule, s->chunk_lines, &c->mark_off};
      got = scan_lines(L, c->text.data() + base, blk, want, c->n_lines, base, &used);
      c->text.resize(base + used);
    }
    const uint64_t b = a + used;
    c->n_lines += got;
    a = b;
    if (c->n_lines >= s->chunk_lines) {
      s->filling = nullptr;
      s->avg_line = (uint32_t)std::max<uint64_t>(16, c->text.size() / c->n_lines + 1);
      const auto td0 = Clock::now();
      if (dispatch(s, c) != 0) return -1;
      g_dbg_ns[7] += (uint64_t)(secs(td0, Clock::now()) * 1e9);
    }
    g_dbg_ns[4] += (uint64_t)(secs(tr1, Clock::now()) * 1e9);
  }
  return s->failed ? -1 : 0;
}

// universal newlines, as Python's open(): "\r\n" and "\r" end a line too.  `p` is rewritten in place; a "\r" that ends a
// block is remembered (carry_cr) so that the "\n" of a "\r\n" split across two calls is not taken for an empty line.
static int write_universal(grim_stream *s, char *p, size_t m) {
  size_t skip = 0;
  if (m && s->carry_cr && p[0] == '\n') skip = 1;
  if (m) s->carry_cr = false;
  if (m && memchr(p, '\r', m)) {
    size_t w = 0;
    for (size_t i = skip; i < m; ++i) {
      if (p[i] == '\r') {
        p[w++] = '\n';
        if (i + 1 < m) {
          if (p[i + 1] == '\n') ++i;
        } else {
          s->carry_cr = true;
        }
      } else {
        p[w++] = p[i];
      }
    }
    m = w;
    skip = 0;
  }
  if (m > skip) return grim_stream_write(s, p + skip, m - skip);
  return s->failed ? -1 : 0;
}

extern "C" int grim_stream_write_text(grim_stream *s, const char *text, uint64_t len) {
  if (!s || (!text && len)) return -1;
  if (!len) return 0;
  if (!s->carry_cr && !memchr(text, '\r', len)) return grim_stream_write(s, text, len);
  std::vector<char> buf(text, text + len);
  return write_universal(s, buf.data(), buf.size());
}

extern "C" int grim_stream_write_file(grim_stream *s, const char *path) {
  if (!s || !path) return -1;
  const int fd = open(path, O_RDONLY);
  if (fd < 0) {
    std::lock_guard<std::mutex> lk(s->mu);
    s->fail(std::string("cannot open ") + path + ": " + strerror(errno));
    return -1;
  }
  std::vector<char> buf(8u << 20);
  int rc = 0;
  for (;;) {
    ssize_t n = read(fd, buf.data(), buf.size());
    if (n < 0) {
      if (errno == EINTR) continue;
      std::lock_guard<std::mutex> lk(s->mu);
      s->fail(std::string("reading ") + path + " failed: " + strerror(errno));
      rc = -1;
      break;
    }
    if (n == 0) break;
    if (write_universal(s, buf.data(), (size_t)n) != 0) {
      rc = -1;
      break;
    }
  }
  close(fd);
  return rc;
}

// closes the chunk that is being filled (a last line without its newline is a line); returns -1 on failure
static int close_filling(grim_stream *s) {
  Chunk *c = s->filling;
  s->filling = nullptr;
  if (!c) return 0;
  if (!c->text.empty() && c->text.back() != '\n') ++c->n_lines;
  if (c->n_lines == 0) {
    std::lock_guard<std::mutex> lk(s->mu);
    c->state = CH_FREE;
    --s->next_index;
    s->cv_slot.notify_all();
    return 0;
  }
  return dispatch(s, c);
}

extern "C" int grim_stream_segment(grim_stream *s, uint64_t next_line_offset) {
  if (!s || s->input_closed) return -1;
  if (close_filling(s) != 0) return -1;
  s->carry_cr = false;
  s->next_line = next_line_offset;
  std::lock_guard<std::mutex> lk(s->mu);
  ++s->cur_segment;
  s->seg_end.push_back({{0, 0, 0, 0, 0, 0, 0}});
  s->seg_set.push_back(0);
  s->seg_begin.push_back({{0, 0, 0, 0, 0, 0, 0}});
  s->seg_begun.push_back(0);
  s->seg_bufs.emplace_back();
  s->seg_first_chunk.push_back(s->next_index);
  s->cv_done.notify_all();  // the segment before this one is closed
  return s->failed ? -1 : 0;
}

// Blocks until every chunk of segment k is committed -- the segment must be closed: a later one opened, or the input
// finished -- and gives the bytes of its piece of each of the seven texts.
extern "C" int grim_stream_segment_wait(grim_stream *s, uint32_t k, uint64_t sizes[7]) {
  if (!s || !sizes) return -1;
  std::unique_lock<std::mutex> lk(s->mu);
  for (;;) {
    if (s->failed || k >= s->seg_end.size()) return -1;
    const bool closed = k < s->cur_segment || s->input_closed;  // (a chunk still being filled is counted by next_index)
    const uint64_t end_chunk = k + 1 < s->seg_first_chunk.size() ? s->seg_first_chunk[k + 1] : s->next_index;
    if (closed && s->next_commit >= end_chunk) break;
    s->cv_done.wait_for(lk, std::chrono::milliseconds(50));
  }
  if (!s->seg_set[k]) {  // a segment without a line
    for (int t = 0; t < 7; ++t) sizes[t] = 0;
    return 0;
  }
  for (int t = 0; t < 7; ++t) sizes[t] = s->seg_end[k][t] - s->seg_begin[k][t];
  return 0;
}

// Placed output: writes segment k's piece of every output file at base[t] (worker threads; returns when the bytes are
// written).  A segment is placed once, after grim_stream_segment_wait returned for it.
extern "C" int grim_stream_segment_place(grim_stream *s, uint32_t k, const uint64_t base[6]) {
  if (!s || !base || !s->placed) return -1;
  std::atomic<int> left{0};
  std::unique_lock<std::mutex> lk(s->mu);
  if (k >= s->seg_bufs.size() || s->failed) return -1;
  std::vector<grim_stream::SegBuf> bufs;
  bufs.swap(s->seg_bufs[k]);
  if (bufs.empty()) return 0;
  left.store((int)bufs.size());
  for (const auto &b : bufs) s->q_place.push_back({b.k, b.p, b.n, base[b.k] + b.rel, &left});
  s->cv_work.notify_all();
  // (a failed stream's workers still drain the queue: `left` reaches zero either way)
  s->cv_place.wait(lk, [&] { return left.load() == 0; });
  return s->failed ? -1 : 0;
}

extern "C" uint32_t grim_stream_n_segments(const grim_stream *s) { return s ? (uint32_t)s->seg_end.size() : 0; }

extern "C" int grim_stream_segment_end(const grim_stream *s, uint32_t k, uint64_t out[7]) {
  if (!s || !out || k >= s->seg_end.size()) return -1;
  // a segment without a line ends where the one before it ended
  int j = (int)k;
  while (j >= 0 && !s->seg_set[j]) --j;
  for (int t = 0; t < 7; ++t) out[t] = j >= 0 ? s->seg_end[j][t] : 0;
  return 0;
}

extern "C" int grim_stream_finish(grim_stream *s) {
  if (!s) return -1;
  if (!s->input_closed) {
    {
      std::lock_guard<std::mutex> lk(s->mu);  // (grim_stream_segment_wait reads it from another thread)
      s->input_closed = true;
    }
    if (close_filling(s) != 0) return -1;
  }
  std::unique_lock<std::mutex> lk(s->mu);
  for (;;) {
    if (s->failed) break;
    if (s->opt.want_records ? s->next_commit >= s->next_index : s->n_done >= s->next_index) break;
    s->cv_done.wait_for(lk, std::chrono::milliseconds(50));
  }
  s->st.wall_s = secs(s->t_open, Clock::now());
  return s->failed ? -1 : 0;
}

extern "C" const char *grim_stream_error(const grim_stream *s) { return s ? s->err.c_str() : "no stream"; }

extern "C" const char *grim_stream_text(grim_stream *s, int which, uint64_t *len) {
  if (!s || which < 0 || which > 6) return nullptr;
  if (!s->joined_ok[which]) {
    size_t tot = 0;
    for (auto &b : s->mem[which]) tot += b.second;
    s->joined[which].clear();
    s->joined[which].reserve(tot);
    for (auto &b : s->mem[which]) {
      s->joined[which].append(b.first, b.second);
      free(b.first);
    }
    s->mem[which].clear();
    s->joined_ok[which] = true;
  }
  if (len) *len = s->joined[which].size();
  return s->joined[which].data();
}

extern "C" int grim_stream_get_stats(const grim_stream *s, grim_stream_stats *out) {
  if (!s || !out) return -1;
  {
    std::lock_guard<std::mutex> lk(const_cast<grim_stream *>(s)->mu);
    *out = s->st;
  }
  out->tokenize_cpu_s = s->tok_ns.load() * 1e-9;
  out->format_cpu_s = s->fmt_ns.load() * 1e-9;
  out->write_cpu_s = s->wr_ns.load() * 1e-9;
  out->unsupported = s->unsupported.size();
  out->bytes_h2d = engine_bytes_moved(nullptr, 0);
  out->bytes_d2h = engine_bytes_moved(nullptr, 1);
  return 0;
}

extern "C" uint64_t grim_stream_n_unsupported(const grim_stream *s) { return s ? s->unsupported.size() : 0; }

extern "C" int grim_stream_unsupported(const grim_stream *s, uint64_t k, uint64_t *line, uint32_t *reason, const char **id, uint32_t *id_len) {
  if (!s || k >= s->unsupported.size()) return -1;
  const auto &u = s->unsupported[k];
  if (line) *line = u.line;
  if (reason) *reason = u.reason;
  if (id) *id = u.id.data();
  if (id_len) *id_len = (uint32_t)u.id.size();
  return 0;
}

extern "C" int grim_stream_next_records(grim_stream *s, grim_stream_records *out) {
  if (!s || !out || !s->opt.want_records) return -1;
  std::unique_lock<std::mutex> lk(s->mu);
  for (;;) {
    if (s->failed) return -1;
    Chunk *c = s->by_index(s->next_record);
    if (c && c->state >= CH_COMMITTED && !c->records_out) {
      c->records_out = true;
      tl_note(c, 8);
      ++s->next_record;
      out->first_line = c->first_line;
      out->n_lines = c->n_lines;
      out->kinds = c->kinds.data();
      out->res = engine_batch_host(c->batch)->res;
      out->rows = c->rows;
      out->chunk = c;
      return 1;
    }
    if (s->input_closed && s->next_record >= s->next_index) return 0;
    const auto tw0 = Clock::now();
    s->cv_rec.wait_for(lk, std::chrono::milliseconds(50));
    g_dbg_ns[6] += (uint64_t)(secs(tw0, Clock::now()) * 1e9);
  }
}

extern "C" int grim_stream_release_records(grim_stream *s, grim_stream_records *rec) {
  if (!s || !rec || !rec->chunk) return -1;
  Chunk *c = (Chunk *)rec->chunk;
  std::lock_guard<std::mutex> lk(s->mu);
  c->records_released = true;
  tl_note(c, 9);
  tl_close(c);
  rec->chunk = nullptr;
  release_if_done(s, c);
  return 0;
}

// Worker threads of a stream when the caller does not say: the cores this process may run on (its affinity mask, not the
// machine's core count) divided by the ranks that share the host (LOCAL_WORLD_SIZE as torchrun exports it, else WORLD_SIZE:
// one node), at most 32.  Eight ranks on one host are then eight streams of cores / 8 threads each, not eight of 32.
extern "C" uint32_t grim_default_threads(void) {
  unsigned cores = 0;
  cpu_set_t set;
  CPU_ZERO(&set);
  if (sched_getaffinity(0, sizeof(set), &set) == 0) cores = (unsigned)CPU_COUNT(&set);
  if (!cores) cores = std::max(1u, std::thread::hardware_concurrency());
  unsigned ranks = 1;
  const char *lw = getenv("LOCAL_WORLD_SIZE");
  if (!lw || atoi(lw) <= 0) lw = getenv("WORLD_SIZE");
  if (lw && atoi(lw) > 0) ranks = (unsigned)atoi(lw);
  unsigned nt = cores / ranks;
  if (nt < 1) nt = 1;
  if (nt > 32) nt = 32;
  return nt;
}

// GRIM_DEBUG_SEGV=1: a SIGSEGV in any of the pipeline's threads prints the faulting thread's native frames (addresses inside
// libgrim_hip.so resolve with addr2line against the same file) before the default action takes over.  Diagnostic aid only.
static void segv_backtrace(int sig, siginfo_t *si, void *) {
  void *frames[48];
  const int n = backtrace(frames, 48);
  char buf[512];
  const int m = snprintf(buf, sizeof(buf), "grim: SIGSEGV at address %p; copy job: k=%llu dst=%llx src=%llx n=%llu nl=%llx nl_size=%llu ticket=%llx np_gen=%llx\n"
                         "native frames of the faulting thread:\n", si ? si->si_addr : nullptr, (unsigned long long)tl_dbg[0], (unsigned long long)tl_dbg[1],
                         (unsigned long long)tl_dbg[2], (unsigned long long)tl_dbg[3], (unsigned long long)tl_dbg[4], (unsigned long long)tl_dbg[5],
                         (unsigned long long)tl_dbg[6], (unsigned long long)tl_dbg[7]);
  (void)!write(2, buf, (size_t)m);
  backtrace_symbols_fd(frames, n, 2);
  signal(sig, SIG_DFL);
  raise(sig);
}

extern "C" grim_stream *grim_stream_open(grim_ctx *ctx, const grim_graph *g, const grim_dict *dict, const grim_params *prm,
                                         const grim_prior_spec *priors, const char *const *pop_names, uint32_t n_pops,
                                         const grim_stream_opts *opts) {
  if (!ctx || !g || !dict || !prm || !priors || !pop_names || !opts || n_pops == 0) return nullptr;
  if (getenv("GRIM_DEBUG_SEGV")) {
    struct sigaction sa;
    memset(&sa, 0, sizeof(sa));
    sa.sa_sigaction = segv_backtrace;
    sa.sa_flags = SA_SIGINFO;
    sigaction(SIGSEGV, &sa, nullptr);
  }
  grim_stream *s = new grim_stream();
  s->ctx = ctx;
  s->graph = g;
  s->prm = *prm;
  s->opt = *opts;
  s->n_pops = n_pops;
  s->next_line = opts->line_offset;
  s->seg_end.push_back({{0, 0, 0, 0, 0, 0, 0}});
  s->seg_set.push_back(0);
  s->seg_begin.push_back({{0, 0, 0, 0, 0, 0, 0}});
  s->seg_begun.push_back(0);
  s->seg_bufs.emplace_back();
  s->seg_first_chunk.push_back(0);
  s->placed = opts->placed != 0;
  memset(&s->st, 0, sizeof(s->st));
  dict_snapshot(dict, s->snap);
  s->races.ps.alpha = priors->alpha;
  s->races.ps.eta = priors->eta;
  s->races.ps.beta = priors->beta;
  s->races.ps.gamma = priors->gamma;
  s->races.ps.delta = priors->delta;
  s->races.ps.unk_mr = priors->unk_mr != 0;
  for (uint32_t i = 0; i < n_pops; ++i) {
    s->races.ps.pops.emplace_back(pop_names[i]);
    s->races.ps.count_by_prob.push_back(priors->count_by_prob ? priors->count_by_prob[i] : 1.0);
    s->fp.pops.emplace_back(pop_names[i]);
  }
  s->fp.snap = &s->snap;
  s->fp.prm = &s->prm;
  s->fp.want_log = opts->want_log != 0;
  if (opts->mask_ids && opts->n_masks) {
    const char *p = opts->mask_ids;
    for (uint32_t i = 0; i < opts->n_masks; ++i) {
      const size_t n = strlen(p);
      s->masks.fixed[std::string(p, n)] = opts->mask_fixed ? opts->mask_fixed[i] : 0;
      p += n + 1;
    }
    s->have_masks = true;
  } else if (opts->mask_ids) {
    s->have_masks = true;  // an empty table: every id is missing
  }
  s->rule.small_ok = (n_pops == 1) && prm->opt_threshold > 1 && !getenv("GRIM_NO_SMALL") && engine_graph_order_bad(g) == 0;
  s->rule.medium_ok = !getenv("GRIM_NO_MEDIUM") && engine_graph_order_bad(g) == 0;
  s->rule.medium_max_cost = getenv("GRIM_MEDIUM_MAXCOST") ? atof(getenv("GRIM_MEDIUM_MAXCOST")) : 0.0;
  s->rule.graph_loci = s->snap.n_loci;
  s->rule.opt_threshold = prm->opt_threshold;
  s->chunk_lines = opts->chunk_lines ? opts->chunk_lines : 131072u;
  s->granule = 1024;
  if (s->chunk_lines < s->granule) s->granule = s->chunk_lines;
  s->depth = opts->depth ? opts->depth : 4u;
  int nt = opts->n_threads;
  if (nt <= 0) nt = (int)grim_default_threads();
  s->n_threads = (uint32_t)nt;
  // row pool of a chunk: bounded; never less than the fixed-stride region of the half-wave kernel plus one subject's
  // worst case and the one-wave kernel's per-wave row blocks
  const uint64_t per = engine_rows_per_subject(prm, n_pops);
  uint64_t rows = opts->rows_per_chunk ? opts->rows_per_chunk : 32ull * s->chunk_lines;
  // (one subject: at most four block grabs of the one-wave kernel, each max(rows needed, 64); more subjects than the pool
  //  holds are what the split-and-rerun is for)
  const uint64_t floor_rows = 2ull * engine_small_stride(prm) * s->chunk_lines + 2 * per + 4 * 64 + 1024;  // staging + its compacted copy
  if (rows < floor_rows && !(opts->rows_exact && opts->rows_per_chunk)) rows = floor_rows;
  if (rows > 0x7FFFFFF0ull) rows = 0x7FFFFFF0ull;
  s->rows_per_chunk = rows;
  // the pool grows when a chunk runs out of rows -- unless the caller set its size (a memory bound is a bound; the tests
  // use it to drive the split-and-rerun path): up to every line's worst case
  s->rows_max = opts->rows_per_chunk ? 0 : std::max<uint64_t>(rows, std::min<uint64_t>(0x7FFFFFF0ull, floor_rows + per * s->chunk_lines));
  // Device tokenizer: lines without a '/' list are parsed on the GPU when their subjects are the half-wave kernel's (one
  // population, five loci, no phase masks); GRIM_DEVICE_TOKENIZER=0 keeps every line on the host (tests hold the two together)
  {
    const char *e = getenv("GRIM_DEVICE_TOKENIZER");
    s->dev_tok = s->rule.small_ok && !s->have_masks && s->snap.n_loci == GRIM_MAXL && !(e && atoi(e) == 0);
    if (s->dev_tok) {
      s->devdict = engine_devdict_create(ctx, &s->snap);
      if (!s->devdict) {
        delete s;
        return nullptr;
      }
    }
  }
  // the batches first: a stream that cannot get its device memory leaves the caller's files alone
  EnginePlan plan0{s->chunk_lines, (uint64_t)s->chunk_lines * 48, s->dev_tok ? (uint64_t)s->chunk_lines * 128 : 0ull};
  for (uint32_t i = 0; i < s->depth; ++i) {
    std::unique_ptr<Chunk> c(new Chunk());
    c->slot_no = (int)i;
    c->batch = engine_batch_create(ctx, g, prm, rows, &plan0);
    if (!c->batch) {
      for (auto &o : s->chunks) grim_batch_free(o->batch);
      engine_devdict_free(s->devdict);
      delete s;
      return nullptr;
    }
    engine_batch_set_dict(c->batch, s->devdict);
    if (opts->timing) grim_batch_set_timing(c->batch, 1);
    s->chunks.push_back(std::move(c));
  }
  // An output file of an earlier run is moved aside and unlinked by a helper thread while the pipeline runs: truncating
  // 400 MB of cached pages costs 30 ms at open(), and ext4 then flushes a truncated-and-rewritten file at close() (another
  // 35 ms) -- more than the pipeline itself needs for a million subjects.  The name it is parked under is this process's
  // own and is never one that exists; when the stream cannot be opened after all, the parked files go back.
  std::vector<std::pair<std::string, std::string>> parked;  // (parked name, original name)
  static std::atomic<unsigned> park_no{0};
  for (int k = 0; k < 6; ++k)
    if (opts->out_path[k]) {
      struct stat sb;
      if (!s->placed && stat(opts->out_path[k], &sb) == 0 && S_ISREG(sb.st_mode) && sb.st_size > (1 << 20)) {
        const std::string old = std::string(opts->out_path[k]) + ".grim_old." + std::to_string((long)getpid()) + "." + std::to_string(park_no++);
        struct stat so;
        if (lstat(old.c_str(), &so) != 0 && errno == ENOENT && rename(opts->out_path[k], old.c_str()) == 0) parked.emplace_back(old, opts->out_path[k]);
      }
      // (placed output: the file is the job's, other ranks write into it too -- whoever made it emptied it)
      s->fd[k] = open(opts->out_path[k], s->placed ? (O_WRONLY | O_CREAT) : (O_WRONLY | O_CREAT | O_TRUNC), 0644);
      if (s->fd[k] < 0) {
        engine_set_error(ctx, (std::string("grim_stream_open: cannot create ") + opts->out_path[k] + ": " + strerror(errno)).c_str());
        for (int j = 0; j < k; ++j)
          if (s->fd[j] >= 0) close(s->fd[j]);
        for (auto &pr : parked) rename(pr.first.c_str(), pr.second.c_str());  // the earlier run's results stay what they were
        for (auto &o : s->chunks) grim_batch_free(o->batch);
        engine_devdict_free(s->devdict);
        delete s;
        return nullptr;
      }
    }
  for (auto &pr : parked) s->stale.push_back(pr.first);
  if (!s->stale.empty())
    s->unlinker = std::thread([s]() {
      for (const std::string &f : s->stale) unlink(f.c_str());
    });
  s->t_open = Clock::now();
  for (uint32_t i = 0; i < s->n_threads; ++i) s->workers.emplace_back(worker_loop, s);
  s->dev_thread = std::thread(device_loop, s);
  s->copy_thread = std::thread(copy_loop, s);
  s->fetch_thread = std::thread(fetch_loop, s);
  {
    // copy helpers of the reader: up to seven, never more than the stream's share of the host's cores leaves room for
    // (measured per 1.07 MB chunk of cold caller memory on 32-thread streams: 53-79 us with three helpers depending on the
    // box, 44-46 us with seven -- and the reader is what bounds the headline on the slower boxes)
    const char *e = getenv("GRIM_COPY_THREADS");
    size_t nc = e ? (size_t)atoi(e) : (s->n_threads >= 24 ? 7u : s->n_threads >= 12 ? 5u : s->n_threads >= 8 ? 3u : s->n_threads >= 4 ? 1u : 0u);
    if (nc > 7) nc = 7;
    s->copy_jobs.resize(COPY_PIECES);
    for (size_t k = 0; k < nc; ++k) s->copiers.emplace_back(copier_loop, s, k);
  }
  return s;
}

extern "C" void grim_stream_free(grim_stream *s) {
  if (!s) return;
  if (getenv("GRIM_DEBUG_STREAM"))
    fprintf(stderr, "grim stream: device thread ms: staging %.3f load %.3f | copy thread ms: wait + stage 2 %.3f fetch %.3f | reader ms: work %.3f "
            "(copy %.3f scan %.3f dispatch %.3f) waiting for a slot %.3f | consumer waiting %.3f over %llu chunks\n", g_dbg_ns[0] / 1e6, g_dbg_ns[1] / 1e6, g_dbg_ns[2] / 1e6,
            g_dbg_ns[3] / 1e6, g_dbg_ns[4] / 1e6, g_dbg_ns[8] / 1e6, g_dbg_ns[9] / 1e6, g_dbg_ns[7] / 1e6, g_dbg_ns[5] / 1e6, g_dbg_ns[6] / 1e6, (unsigned long long)s->st.chunks);
  if (getenv("GRIM_DEBUG_STREAM") && g_tl_n.load()) {
    static const char *nm[10] = {"", "fill", "tokenise", "-> device thread", "load + launch", "-> kernels done", "stage 2 + export issued", "-> on the host", "-> consumer", "consumer"};
    fprintf(stderr, "grim stream: a chunk's way, us per chunk (%llu chunks):", (unsigned long long)g_tl_n.load());
    for (int k = 1; k < 10; ++k) fprintf(stderr, " %s %.1f |", nm[k], g_tl_ns[k] / 1e3 / (double)g_tl_n.load());
    fprintf(stderr, "\n");
  }
  {
    std::lock_guard<std::mutex> lk(s->mu);
    s->stop = true;
    s->cv_work.notify_all();
    s->cv_dev.notify_all();
    s->cv_copy.notify_all();
    s->cv_fetch.notify_all();
    s->cv_slot.notify_all();
  }
  {
    std::lock_guard<std::mutex> lk(s->copy_mu);
    s->copy_stop = true;
  }
  s->cv_copyjob.notify_all();
  for (auto &t : s->copiers) t.join();
  for (auto &t : s->workers) t.join();
  if (s->dev_thread.joinable()) s->dev_thread.join();
  if (s->copy_thread.joinable()) s->copy_thread.join();
  if (s->fetch_thread.joinable()) s->fetch_thread.join();
  if (s->unlinker.joinable()) s->unlinker.join();
  for (auto &c : s->chunks) {
    engine_batch_set_dict(c->batch, nullptr);
    engine_batch_recycle(c->batch);  // (waits for the stream: nothing uses the dictionary any more)
  }
  engine_devdict_free(s->devdict);
  for (int k = 0; k < 6; ++k)
    if (s->fd[k] >= 0) close(s->fd[k]);
  for (int k = 0; k < 7; ++k)
    for (auto &b : s->mem[k]) free(b.first);
  for (auto &v : s->seg_bufs)
    for (auto &b : v) free(b.p);  // segments that were never placed (a failed job)
  delete s;
}
