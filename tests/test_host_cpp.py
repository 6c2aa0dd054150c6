"""The library's C++ host helpers (allele dictionary, tokenizer, float formatter) against the Python
host code that restates the same reference rules.  No GPU needed."""
import math
import os
import struct

import numpy as np
import pytest

import harness
import synth


def test_float_text_equals_cpython_repr():
    from grim import _native as nat

    rng = np.random.default_rng(3)
    vals = [0.0, 1.0, 100.0, 1e16, 1e15, 123456789012345.6, 1e-4, 1e-5, 9.999e-5, 0.00012345, 5e-324, 1.7976931348623157e308,
            8.838563003520004e-17, 3.2495135534580065e-15, 0.1, 1 / 3, 2.5e-10, 1e22, 1.5e16, 12345678.9, 6e-05, 0.5]
    vals += list(np.exp(rng.uniform(-60, 5, 3000)))
    vals += [struct.unpack("<d", struct.pack("<Q", int(x)))[0] for x in rng.integers(1, 0x7FEFFFFFFFFFFFFF, 3000)]
    for v in vals:
        if math.isfinite(v):
            assert nat.format_double(v) == repr(float(v)), v
            assert nat.format_double(-v) == repr(float(-v)), v


def _decode(g, subj, toks, parsed=None, lines_of=None):
    """subject record -> structure of allele NAMES (alleles outside the dictionary have per-subject ids: their text
    comes from the parsed block)"""
    out = []
    for si, s in enumerate(subj):
        off = int(s["tok_off"])
        pos = []
        for k in range(int(s["n_loci"])):
            sides = []
            for side in range(2):
                n = int(s["cnt"][k][side])
                sides.append(([parsed.allele(lines_of[si], int(s["slot"][k]), int(t)) for t in toks[off: off + n]], int(s["wid"][k][side])))
                off += n
            pos.append((int(s["slot"][k]), sides))
        out.append((int(s["n_loci"]), int(s["pad"][0]), pos))
    return out


@pytest.mark.parametrize("scenario", ["cau_edge", "cau_mixed", "pop4_edge", "pop4_mixed", "cau_filter", "cau_planc", "cau_irregular"])
def test_cpp_tokenizer_equals_python_tokenizer(scenario):
    from grim import _native as nat
    from grim.imputation import impute as I
    from grim.imputation.networkx_graph import Graph
    from grim.run_impute_def import load_config

    gname, conf, lines, exp, elog, em = harness.golden(scenario)
    work = harness.ensure_graph(gname)
    conf2, cpath = harness._write_inputs(work, conf, lines, "tok_" + scenario)
    cwd = os.getcwd()
    os.chdir(work)
    try:
        cfg, _ = load_config(cpath)
        g = Graph(cfg).build_graph(cfg["node_file"], cfg["top_links_file"], cfg["edges_file"])
        imp = I.Imputation(g, cfg)
    finally:
        os.chdir(cwd)
    extra = ["", "X", "onlyid,", "a,b,c", "Z1,A*01:01+A*02:01^^B*07:02+B*08:01,CAU,CAU", "Z2,A*01:01+B*07:02,CAU,CAU",
             "Z3,A*01:01+A*02:01^A*03:01+A*11:01,CAU,CAU", "Z4,Q*01:01+Q*01:02,CAU,CAU", "Z5,+^A*01:01+A*02:01,CAU,CAU",
             "Z6, ,CAU,CAU", "Z7,A*01:01+A*02:01+A*03:01^B*07:02+B*08:01,CAU,CAU,extra", "Z8%A*01:01+A*02:01%CAU%CAU   "]
    all_lines = lines + extra
    for planb in (True, False):
        parsed = nat.Parsed(g.adict, ("\n".join(all_lines) + "\n").encode(), planb)
        kinds = parsed.kinds()
        dev = parsed.dev_index()
        subj = parsed.subjects()
        toks = parsed.tokens()
        races = parsed.races()
        assert parsed.n_lines == len(all_lines)
        py_kinds, py_recs, py_ids, py_races = [], [], [], []
        for line in all_lines:
            line = line.rstrip()
            sid = None
            try:
                parts = line.split(",") if "," in line else line.split("%")
                sid = parts[0]
                gl = parts[1]
                r1 = r2 = None
                if len(parts) > 2:
                    r1, r2 = parts[2], parts[3]
                kind, payload = imp._tokenise(gl, planb)
                py_kinds.append(kind)
                if kind == I._DEV:
                    py_recs.append(payload)
                    py_races.append((r1 or "", r2 or ""))
            except Exception:
                py_kinds.append(I._PROBLEM_RAW)
            py_ids.append(sid)
        assert list(kinds) == py_kinds
        assert [parsed.subject_id(i) for i in range(len(all_lines)) if py_ids[i] is not None] == [x for x in py_ids if x is not None]
        lines_of = [j for j in range(len(all_lines)) if dev[j] >= 0]
        got = _decode(g, subj, toks, parsed, lines_of)
        assert len(got) == len(py_recs)
        for a, (n, slots, same, pos) in zip(got, py_recs):
            exp_pos = [(slots[k], [([g.adict.name(slots[k], t) for t in pos[k][s][0]], pos[k][s][1]) for s in range(2)]) for k in range(n)]
            assert a == (n, same, exp_pos)
        assert [races[int(s["prior_idx"])] for s in subj] == py_races
        parsed.close()


def test_cpp_prior_matrix_bit_identical_to_python():
    """grim_prior_matrix (C++ restatement of calc_priority_matrix, impute.py:1844-1924) against the Python/numpy
    version the oracle pins (tests/test_host_logic.py): every entry bit for bit."""
    from grim import _native as nat
    from grim.imputation.impute import Imputation

    rng = np.random.default_rng(11)
    pops = ["CAU", "AFA", "HIS", "API", "NAM"]
    for trial in range(40):
        P = int(rng.integers(1, 6))
        pp = pops[:P]
        imp = Imputation.__new__(Imputation)
        imp.populations = pp
        imp.unk_priors = "MR" if trial % 2 else "SR"
        imp.count_by_prob = np.ones(P) if trial % 3 else rng.uniform(0.5, 3.0, P)
        pri = {k: float(v) for k, v in zip(("alpha", "eta", "beta", "gamma", "delta"), rng.uniform(0, 1, 5))}
        if trial % 4 == 0:
            pri = {"alpha": 0.4999999, "eta": 0, "beta": 1e-7, "gamma": 1e-7, "delta": 0.4999999}
        ps, keep = nat.prior_spec(pri, imp.unk_priors, imp.count_by_prob)
        names = pp + ["UNK", "", "XXX"]
        for _ in range(30):
            def race():
                k = int(rng.integers(1, 4))
                return ";".join(names[int(rng.integers(0, len(names)))] for _ in range(k))
            r1, r2 = race(), race()
            if rng.random() < 0.1:
                r1 = r2 = ""
            with np.errstate(all="ignore"):
                a = imp._prior_matrix(r1, r2, pri)
            b = nat.prior_matrix(ps, pp, r1, r2)
            assert np.array_equal(np.asarray(a, dtype=np.float64).view(np.uint64), b.view(np.uint64)), (pp, r1, r2)


def test_fast_tokenizer_path_equals_general_path(monkeypatch):
    """the single-pass fast path for regular GL strings against the general path (GRIM_NO_FAST_TOKENIZER) on seeded
    subjects with the fuzzer's mutations: same kinds, same subject records, same tokens, byte for byte"""
    import sys

    from grim import _native as nat
    from grim.imputation.networkx_graph import Graph
    from grim.run_impute_def import load_config

    sys.path.insert(0, os.path.join(harness.ROOT, "tools"))
    import fuzz

    work = harness.ensure_graph("pop4")
    cwd = os.getcwd()
    os.chdir(work)
    try:
        cfg, _ = load_config("graph_conf.json")
        g = Graph(cfg).build_graph(cfg["node_file"], cfg["top_links_file"], cfg["edges_file"])
    finally:
        os.chdir(cwd)
    rows = synth.read_freqs(synth.CAU_FREQS)
    pops = harness.POPS["pop4"]
    rng = np.random.default_rng(12)
    gen = synth.SubjectGen(rows, 99, pops=pops)
    lines = gen.full(3000) + gen.mixed(6000, amb=0.5, miss=0.3, recomb=0.3)
    lines = [fuzz.mutate(l, rng, gen.by_locus) if rng.random() < 0.3 else l for l in lines]
    text = ("\n".join(lines) + "\n").encode()
    out = []
    for no_fast in (False, True):
        if no_fast:
            monkeypatch.setenv("GRIM_NO_FAST_TOKENIZER", "1")
        else:
            monkeypatch.delenv("GRIM_NO_FAST_TOKENIZER", raising=False)
        parsed = nat.Parsed(g.adict, text, True)
        subj = parsed.subjects().copy()
        races = parsed.races()
        pairs = [races[int(i)] for i in subj["prior_idx"]]  # the numbering of race pairs depends on which thread met one first
        subj["prior_idx"] = 0
        out.append((bytes(parsed.kinds()), subj.tobytes(), parsed.tokens().tobytes(), list(parsed.dev_index()), pairs))
        parsed.close()
    assert out[0][0] == out[1][0]
    assert out[0][3] == out[1][3]
    assert out[0][1] == out[1][1]
    assert out[0][2] == out[1][2]
    assert out[0][4] == out[1][4]
    assert out[0][0].count(bytes([nat.K_DEVICE])) > 8000  # most lines are device subjects (the fast path's clientele)
