"""The streaming pipeline of the library (grim_stream_*, csrc/grim_stream.cpp) on the GPU: chunking, ranges, the bounded
row pool with split-and-rerun, records mode, and BASELINE config 3 (1 M subjects) through properties + an oracle slice.
Everything goes through the C-ABI (ctypes)."""
import os

import numpy as np
import pytest

import harness
import synth

pytestmark = pytest.mark.gpu


def _imp(gname, conf):
    from grim.imputation.impute import Imputation
    from grim.imputation.networkx_graph import Graph
    from grim.run_impute_def import load_config

    work = harness.ensure_graph(gname)
    conf2, cpath = harness._write_inputs(work, conf, [], "stream_cfg")
    cwd = os.getcwd()
    os.chdir(work)
    try:
        cfg, _ = load_config(cpath)
        g = harness._graph_cache.get(gname)
        if g is None:
            g = Graph(cfg).build_graph(cfg["node_file"], cfg["top_links_file"], cfg["edges_file"])
            harness._graph_cache[gname] = g
        imp = Imputation(g, cfg)  # (reads pops_count_file, a path relative to the work directory)
    finally:
        os.chdir(cwd)
    imp.quiet = True
    return imp, cfg


def _stream_texts(imp, cfg, lines, em_mr=False, ctx=None, step=777, borrowed=False, **kw):
    """the six texts of `lines` through a stream opened with explicit options (chunk size, row pool, threads)"""
    from grim import _native as nat

    params = imp._params(cfg, cfg["planb"], em_mr, False)
    ps, keep = nat.prior_spec(cfg["priority"], imp.unk_priors, imp.count_by_prob)
    ctx = ctx or nat.default_context(None)
    st = nat.Stream(ctx, imp.netGraph.device(ctx), imp.netGraph.adict, params, ps, imp.populations, **kw)
    try:
        data = ("\n".join(lines) + "\n").encode() if lines else b""
        # feed in awkward pieces: lines straddle the calls
        for a in range(0, len(data), step):
            st.write(data[a:a + step], borrowed=borrowed)
        st.finish()
        texts = {key: st.text(k) for k, key in enumerate(nat.TEXT_KEYS)}
        stats = st.stats()
        return texts, stats, st.unsupported()
    finally:
        st.close()


@pytest.mark.parametrize("scenario", ["cau_mixed", "pop4_mixed", "cau_edge", "pop4_planc_rerun", "cau_scan30", "cau_mr_res1000"])
def test_small_chunks_and_ranges_equal_golden(scenario):
    """64-line chunks (several chunks in flight, several ranges per chunk on 4 threads): same bytes as the reference"""
    gname, conf, lines, exp, elog, em = harness.golden(scenario)
    imp, cfg = _imp(gname, conf)
    texts, stats, unsup = _stream_texts(imp, cfg, lines, em_mr=em, chunk_lines=64, n_threads=4, depth=3)
    assert not unsup
    assert stats.chunks == (len(lines) + 63) // 64
    for k in exp:
        if (k in ("umug", "umug_pops") and not cfg["output_MUUG"]) or (k in ("pmug", "pmug_pops") and not cfg["output_haplotypes"]):
            continue
        assert texts[k] == exp[k], "%s: %s differs from the reference output" % (scenario, k)


@pytest.mark.parametrize("scenario", ["cau_mixed", "pop4_mixed", "cau_mr_res1000"])
def test_row_pool_overflow_splits_and_reruns(scenario):
    """a row pool far too small for the chunk: the chunk is halved and run again until the parts fit; same bytes"""
    gname, conf, lines, exp, elog, em = harness.golden(scenario)
    imp, cfg = _imp(gname, conf)
    pool = 2600 if scenario == "cau_mr_res1000" else 700  # one subject fits (2 002 / 52 rows at most, plus block slack), the chunk does not
    texts, stats, unsup = _stream_texts(imp, cfg, lines, em_mr=em, chunk_lines=256, rows_per_chunk=pool, rows_exact=True, n_threads=2)
    assert stats.reruns > 0, "the pool was meant to overflow"
    for k in exp:
        assert texts[k] == exp[k], "%s: %s differs after split-and-rerun" % (scenario, k)


def test_block_entry_points_equal_stream():
    """grim_tokenize -> grim_batch_upload/run/results -> grim_format (one batch, caller's thread) against the stream"""
    rows = synth.read_freqs(synth.CAU_FREQS)
    conf = harness.base_conf(harness.POPS["pop4"])
    conf["UNK_priors"] = "MR"
    lines = synth.SubjectGen(rows, 41, pops=harness.POPS["pop4"]).mixed(400) + synth.edge_cases("AFA") + synth.plan_c_cases("HIS")
    imp, cfg = _imp("pop4", conf)
    a = imp.impute_lines_block(lines, cfg)
    b = imp.impute_lines(lines, cfg)
    for k in b:
        assert a[k] == b[k], k


def test_pair_pool_grows_and_the_run_repeats(monkeypatch):
    """a pair pool far too small for the batch (GRIM_PAIR_POOL): the stream and the block entry points (grim_batch_run's
    own retry) size it for what the run asked for, run again, and produce what a roomy pool produces"""
    rows = synth.read_freqs(synth.CAU_FREQS)
    conf = harness.base_conf(harness.POPS["pop4"])
    conf["UNK_priors"] = "MR"
    lines = synth.SubjectGen(rows, 77, pops=harness.POPS["pop4"]).mixed(3000)
    imp, cfg = _imp("pop4", conf)
    monkeypatch.delenv("GRIM_PAIR_POOL", raising=False)
    roomy = imp.impute_lines(lines, cfg)
    monkeypatch.setenv("GRIM_PAIR_POOL", "20000")
    tight = imp.impute_lines(lines, cfg)
    block = imp.impute_lines_block(lines, cfg)
    for k in roomy:
        assert roomy[k] == tight[k], k
        assert roomy[k] == block[k], k


def test_records_mode_matches_texts():
    """want_records: the raw result records of every chunk, in input order, while the texts are built as well"""
    import threading

    from grim import _native as nat

    rows = synth.read_freqs(synth.CAU_FREQS)
    lines = synth.SubjectGen(rows, 42).full(3000) + synth.SubjectGen(rows, 43).mixed(500)
    conf = harness.base_conf(["CAU"])
    imp, cfg = _imp("cau", conf)
    params = imp._params(cfg, cfg["planb"], False, False)
    ps, keep = nat.prior_spec(cfg["priority"], imp.unk_priors, imp.count_by_prob)
    ctx = nat.default_context(None)
    st = nat.Stream(ctx, imp.netGraph.device(ctx), imp.netGraph.adict, params, ps, imp.populations, want_records=True,
                    chunk_lines=512, depth=2, n_threads=3)
    data = ("\n".join(lines) + "\n").encode()
    err = []

    def feed():
        try:
            st.write(data)
            st.finish()
        except Exception as e:  # pragma: no cover
            err.append(e)

    th = threading.Thread(target=feed)
    th.start()
    n_lines = n_rows = n_pairs = 0
    first = []
    while True:
        rec = st.next_records()
        if rec is None:
            break
        first_line, kinds, res, rows_addr, handle = rec
        first.append(first_line)
        dev = kinds == nat.K_DEVICE
        n_lines += len(kinds)
        n_rows += int(res["n_rows"][dev][:, nat.T_PMUG].sum())
        n_pairs += int((res["status"][dev] == nat.ST_OK).sum())
        st.release(handle)
    th.join()
    assert not err
    assert first == list(range(0, len(lines), 512))
    assert n_lines == len(lines)
    pmug = st.text(2)
    assert n_rows == len(pmug.splitlines())
    assert n_pairs == len({l.split(",")[0] for l in st.text(0).splitlines()})
    st.close()


def test_input_file_with_crlf_and_no_final_newline(tmp_path):
    """grim_stream_write_file: universal newlines as Python's open(), last line without '\\n'"""
    rows = synth.read_freqs(synth.CAU_FREQS)
    lines = synth.SubjectGen(rows, 44).mixed(300)
    conf = harness.base_conf(["CAU"])
    imp, cfg = _imp("cau", conf)
    ref = imp.impute_lines(lines, cfg)
    p = tmp_path / "in.csv"
    p.write_bytes("\r\n".join(lines[:150]).encode() + b"\r" + "\r".join(lines[150:]).encode())
    cfg2 = dict(cfg)
    cfg2["imputation_input_file"] = str(p)
    out = {}
    for key, path_key, flag in imp._OUT_FILES:
        cfg2[path_key] = str(tmp_path / (key + ".txt"))
    imp.impute_file(cfg2)
    for key, path_key, flag in imp._OUT_FILES:
        assert open(cfg2[path_key]).read() == ref[key], key


def test_config3_one_million_subjects_properties_and_oracle_slice():
    """BASELINE configs[2]: CAU 5-locus, 1 M fully typed subjects, seed 1 (one GPU's worth here: the graph is replicated
    and subjects are independent, so a shard is the same computation).  File -> file through impute_file."""
    rows = synth.read_freqs(synth.CAU_FREQS)
    n = 1_000_000
    lines = synth.SubjectGen(rows, 1).full_fast(n)
    conf = harness.base_conf(["CAU"])
    got, glog, imp = harness.run_product("cau", conf, lines, tag="c3", quiet=True)
    assert imp.last_stats["lines"] == n and imp.last_stats["chunks"] >= 7
    assert got["miss"] == "" and got["problem"] == ""
    umug = got["umug"].splitlines()
    assert len(umug) == n
    # one MUUG per subject in input order, rank 0
    k = 0
    for i in range(0, n, 9973):
        f = umug[i].split(",")
        assert f[0] == "S%d" % i and f[3] == "0"
        k += 1
    # checksum of checksums over permutation: a permuted slice gives the same rows for the same ids
    perm = np.random.default_rng(7).permutation(50_000)
    sub = [lines[i] for i in perm]
    got2, _, _ = harness.run_product("cau", conf, sub, tag="c3p", quiet=True)
    want = {l.split(",")[0]: l for l in umug[:50_000]}
    for l in got2["umug"].splitlines():
        assert want[l.split(",")[0]] == l
    # phased rows: ranked, and they add up to the MUUG when all are listed
    by = {}
    for l in got["pmug"].splitlines()[:400_000]:
        f = l.split(",")
        by.setdefault(f[0], []).append(float(f[2]))
    for i in range(0, 60_000, 7):
        ps = by.get("S%d" % i)
        if ps is None:
            continue
        assert ps == sorted(ps, reverse=True)
        u = float(umug[i].split(",")[2])
        if len(ps) < 10:
            assert abs(sum(ps) - u) <= 1e-12 * u
    # an oracle slice from the middle of the file (all six files)
    lo, hi = 500_000, 501_500
    exp, _ = harness.run_oracle("cau", conf, lines[lo:hi], tag="c3_orc")
    ids = {l.split(",")[0] for l in lines[lo:hi]}
    for key in ("umug", "umug_pops", "pmug", "pmug_pops"):
        mine = [l for l in got[key].splitlines() if l.split(",", 1)[0] in ids]
        assert mine == exp[key].splitlines(), key


def test_device_tokenizer_equals_host_tokenizer(monkeypatch):
    """GL strings parsed ON THE DEVICE (lines without a '/' list on a one-population graph: grim_tokdev.h) against the
    host tokenizer (GRIM_DEVICE_TOKENIZER=0) and against the oracle: fully typed subjects with the fuzzer's mutations --
    alleles the graph has never seen (overlay ids), 'g' / 'L' suffixes, UUUU loci, homozygous loci, missing loci, malformed
    and truncated lines, '/' lists -- so that the device takes most lines, hands some back (the fix-up pass) and never sees
    the rest.  Small chunks, ragged pieces, several ranges per chunk; then one big chunk; then records mode."""
    import sys

    from grim import _native as nat

    sys.path.insert(0, os.path.join(harness.ROOT, "tools"))
    import fuzz

    rows = synth.read_freqs(synth.CAU_FREQS)
    rng = np.random.default_rng(21)
    gen = synth.SubjectGen(rows, 98)
    lines = gen.full(5000) + gen.mixed(800, amb=0.3, miss=0.3, recomb=0.2)
    lines = [fuzz.mutate(l, rng, gen.by_locus) if rng.random() < 0.25 else l for l in lines]
    # loci out of sorted order (gl2haps sorts: same subject), a locus twice, a long allele name, '%' separators
    swapped = []
    for l in gen.full(40):
        f = l.split(",")
        loci = f[1].split("^")
        loci[1], loci[3] = loci[3], loci[1]
        swapped.append(",".join([f[0] + "s", "^".join(loci)] + f[2:]))
    twice = ["T%d%s" % (i, l[l.index(","):].replace("^C*", "^A*", 1)) for i, l in enumerate(gen.full(10))]  # (ids of their own)
    longname = [l.replace("+", ":01:01:01:01:01:01:01+", 1) for l in gen.full(10)]
    percent = [l.replace(",", "%") for l in gen.full(30)]
    lines += swapped + twice + longname + percent
    conf = harness.base_conf(["CAU"])
    imp, cfg = _imp("cau", conf)
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("GRIM_DEVICE_TOKENIZER", mode)
        # the lines that name a locus twice are REPORTED (reason 8), by either tokenizer setting, with their line numbers
        twice_at = [(i, l.split(",")[0], 8) for i, l in enumerate(lines) if l in set(twice)]
        out[mode, "small"], st_small, unsup = _stream_texts(imp, cfg, lines, chunk_lines=700, n_threads=3, depth=3)
        assert sorted(unsup) == twice_at
        out[mode, "big"], st_big, unsup = _stream_texts(imp, cfg, lines, n_threads=5)
        assert sorted(unsup) == twice_at
    monkeypatch.delenv("GRIM_DEVICE_TOKENIZER", raising=False)
    for k in nat.TEXT_KEYS:
        assert out["1", "small"][k] == out["0", "small"][k], k
        assert out["1", "big"][k] == out["0", "big"][k], k
        assert out["1", "big"][k] == out["1", "small"][k], k
    # against the oracle on ALL lines: the reported subjects (the oracle, like the reference, answers them: gl2haps pairs the
    # entries by index) are the only lines missing from the product's files
    got, _, unsup = _stream_texts(imp, cfg, lines, chunk_lines=900, n_threads=4)
    assert sorted(sid for _, sid, _ in unsup) == sorted(l.split(",")[0] for l in twice)
    exp, _ = harness.run_oracle("cau", conf, lines, tag="devtok_orc")
    exp = harness.drop_subjects(exp, [sid for _, sid, _ in unsup])
    for k in exp:
        assert got[k] == exp[k], k
    # the device really took part: a stream of only regular lines has no host-tokenised subject at all
    clean = gen.full(3000)
    t1, st1, _ = _stream_texts(imp, cfg, clean, chunk_lines=1000, n_threads=2, timing=True)
    monkeypatch.setenv("GRIM_DEVICE_TOKENIZER", "0")
    t0, st0, _ = _stream_texts(imp, cfg, clean, chunk_lines=1000, n_threads=2, timing=True)
    monkeypatch.delenv("GRIM_DEVICE_TOKENIZER", raising=False)
    for k in nat.TEXT_KEYS:
        assert t1[k] == t0[k], k
    assert st1.bytes_h2d > 0 and st1.subjects == st0.subjects == 3000


def test_one_big_write_whose_pieces_divide_evenly():
    """The reader copies a big write in up to 64 pieces (grim_stream_write): a block whose size / 64 is a multiple of the
    piece granule plus a remainder used to be cut into 65 -- one more than the job array holds (a crash in the sharded
    driver, whose segments happened to have such sizes).  One write of exactly 64 * 64 * 800 + 5 bytes against the same
    text fed in small pieces."""
    from grim import _native as nat

    rows = synth.read_freqs(synth.CAU_FREQS)
    lines = synth.SubjectGen(rows, 77).full(40000)
    want = 64 * 64 * 800 + 5
    text, n = [], 0
    for l in lines:
        if n + len(l) + 1 > want - 200:
            break
        text.append(l)
        n += len(l) + 1
    pad = want - n - 1 - len(lines[len(text)])  # the last line's id grows by what is missing
    last = lines[len(text)]
    text.append("X" * pad + last)
    data = ("\n".join(text) + "\n").encode()
    assert len(data) == want
    conf = harness.base_conf(["CAU"])
    imp, cfg = _imp("cau", conf)
    params = imp._params(cfg, cfg["planb"], False, False)
    ps, keep = nat.prior_spec(cfg["priority"], imp.unk_priors, imp.count_by_prob)
    ctx = nat.default_context(None)
    out = []
    for step in (len(data), 4099):
        st = nat.Stream(ctx, imp.netGraph.device(ctx), imp.netGraph.adict, params, ps, imp.populations, n_threads=16)
        try:
            for a in range(0, len(data), step):
                st.write(data[a:a + step])
            st.finish()
            out.append({key: st.text(k) for k, key in enumerate(nat.TEXT_KEYS)})
            assert not st.unsupported()
        finally:
            st.close()
    for k in out[0]:
        assert out[0][k] == out[1][k], k
    assert out[0]["umug"].count("\n") == len(text)


def test_results_come_down_on_their_own_sdma_engine(monkeypatch):
    """A stream's results leave HBM on an SDMA engine the library names itself (csrc/grim_sdma.h: ROCr's
    hsa_amd_memory_async_copy_on_engine; the HIP runtime would put uploads and downloads on engine 0 both).  The default
    context of an MI355X box gets one (a bit > 1: not the uploads' engine), and the two other ways down -- the copy
    kernel of round 3 (GRIM_EXPORT=kernel) and hipMemcpyAsync (GRIM_EXPORT=memcpy) -- must give the same six files, small
    chunks and one big chunk, and the same records."""
    from grim import _native as nat

    rows = synth.read_freqs(synth.CAU_FREQS)
    gen = synth.SubjectGen(rows, 4242)
    lines = gen.full(6000) + gen.mixed(600, amb=0.3, miss=0.3, recomb=0.2)
    conf = harness.base_conf(["CAU"])
    imp, cfg = _imp("cau", conf)
    # (a box whose ROCr offers no engine besides the uploads' falls back to the copy kernel -- not an error of the product:
    #  the other two ways are still held against each other there, and the test says so)
    have_sdma = nat.default_context(None).export_engine() > 1
    base_small, _, _ = _stream_texts(imp, cfg, lines, chunk_lines=900, n_threads=3, depth=3)
    base_big, _, _ = _stream_texts(imp, cfg, lines, n_threads=4)
    for mode, engine in (("kernel", 0), ("memcpy", -1)) + ((("sdma", None),) if have_sdma else ()):
        monkeypatch.setenv("GRIM_EXPORT", mode)
        ctx = nat.Context(0)
        try:
            assert ctx.export_engine() == engine if engine is not None else ctx.export_engine() > 1
            small, _, _ = _stream_texts(imp, cfg, lines, ctx=ctx, chunk_lines=900, n_threads=3, depth=3)
            big, _, _ = _stream_texts(imp, cfg, lines, ctx=ctx, n_threads=4)
        finally:
            dg = imp.netGraph._dev.pop(id(ctx), None)  # the graph's HBM copy on this context goes before the context does
            if dg is not None:
                dg.close()
            ctx.close()
        for k in nat.TEXT_KEYS:
            assert small[k] == base_small[k], (mode, k)
            assert big[k] == base_big[k], (mode, k)
    monkeypatch.delenv("GRIM_EXPORT", raising=False)
    for k in nat.TEXT_KEYS:
        assert base_small[k] == base_big[k], k
    if not have_sdma:
        pytest.skip("no SDMA engine besides the uploads' on this box: copy kernel and hipMemcpyAsync compared, SDMA path not run")


def test_borrowed_input_equals_copied_input():
    """grim_stream_write_borrowed: chunks that begin inside a lent buffer read their lines where they lie (the reader only
    finds the line ends).  One big write, two writes cut in the middle of a line, and writes of 300 001 bytes (every one big
    enough for views, every boundary inside a line, so that a chunk starts as a view and is completed from the next buffer)
    must give the six texts of the copying path fed in 777-byte pieces -- ids, which the formatter reads from the chunk's
    text, included."""
    from grim import _native as nat

    rows = synth.read_freqs(synth.CAU_FREQS)
    gen = synth.SubjectGen(rows, 777)
    lines = gen.full(30000) + gen.mixed(1500, amb=0.3, miss=0.3, recomb=0.2) + gen.full(2500)
    conf = harness.base_conf(["CAU"])
    imp, cfg = _imp("cau", conf)
    n_bytes = sum(len(l) + 1 for l in lines)
    assert n_bytes > 3000000
    base, _, _ = _stream_texts(imp, cfg, lines, chunk_lines=4000, n_threads=4)
    for step in (n_bytes, n_bytes // 2 + 13, 300001):
        got, st, _ = _stream_texts(imp, cfg, lines, chunk_lines=4000, n_threads=4, step=step, borrowed=True)
        assert st.lines == len(lines)
        for k in nat.TEXT_KEYS:
            assert got[k] == base[k], (step, k)
    # the default chunk size (one chunk for everything) and a depth of two
    got, _, _ = _stream_texts(imp, cfg, lines, n_threads=5, depth=2, step=n_bytes, borrowed=True)
    for k in nat.TEXT_KEYS:
        assert got[k] == base[k], k
    # a four-population graph: every line is tokenised by the HOST's threads, straight from the lent buffer
    pops = harness.POPS["pop4"]
    lines4 = synth.SubjectGen(rows, 5, pops=pops).mixed(4000)
    conf4 = harness.base_conf(pops)
    conf4["UNK_priors"] = "MR"
    imp4, cfg4 = _imp("pop4", conf4)
    n4 = sum(len(l) + 1 for l in lines4)
    assert n4 > 400000
    base4, _, _ = _stream_texts(imp4, cfg4, lines4, chunk_lines=1000, n_threads=4)
    for step in (n4, n4 // 2 + 7):
        got4, _, _ = _stream_texts(imp4, cfg4, lines4, chunk_lines=1000, n_threads=4, step=step, borrowed=True)
        for k in nat.TEXT_KEYS:
            assert got4[k] == base4[k], (step, k)


def test_big_input_files_with_every_line_end(tmp_path, monkeypatch):
    """grim_stream_write_file on files of 4.5 MB (several 8 MB read blocks would be more; several chunks are): the same 40 000
    lines with '\\n' ends and no final newline, with '\\r\\n' ends throughout, and with '\\n' ends but '\\r\\n' in the last tenth
    and a lone '\\r' at the very end -- each against impute_lines on the same lines.  (Written for a variant of write_file that
    mapped the file and lent the mapping to the stream until the first '\\r': correct, no faster -- file -> file is bound by
    formatting and writing -- and dropped; the test stays.)"""
    rows = synth.read_freqs(synth.CAU_FREQS)
    gen = synth.SubjectGen(rows, 91)
    lines = gen.full(36000) + gen.mixed(4000, amb=0.3, miss=0.3, recomb=0.2)
    conf = harness.base_conf(["CAU"])
    imp, cfg = _imp("cau", conf)
    ref = imp.impute_lines(lines, cfg)
    n = len(lines)
    variants = {
        "lf": "\n".join(lines).encode(),
        "crlf": "\r\n".join(lines).encode() + b"\r\n",
        "late_cr": "\n".join(lines[:n - n // 10]).encode() + b"\n" + "\r\n".join(lines[n - n // 10:]).encode() + b"\r",
    }
    for name, data in variants.items():
        assert len(data) > (4 << 20)
        p = tmp_path / (name + ".csv")
        p.write_bytes(data)
        cfg2 = dict(cfg)
        cfg2["imputation_input_file"] = str(p)
        for key, path_key, flag in imp._OUT_FILES:
            cfg2[path_key] = str(tmp_path / (name + "_" + key + ".txt"))
        monkeypatch.setenv("GRIM_CHUNK_LINES", "6000")  # several chunks, most of them views
        imp.impute_file(cfg2)
        for key, path_key, flag in imp._OUT_FILES:
            assert open(cfg2[path_key]).read() == ref[key], (name, key)
