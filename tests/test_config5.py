"""BASELINE configs[4] stand-in (tools/wmda_scale.py): a synthetic WMDA-scale multi-population graph and high-ambiguity
subjects (8 / 16 alternatives per locus and side, options threshold 1e6, 100 haplotypes in phase).
 * a 20 000-haplotype variant of the same recipe against the oracle (the oracle's Python dicts of the full graph would
   take minutes and gigabytes);
 * the full graph (300 000 haplotypes: ~1.1 M nodes, 8.6 M top links, three populations) through size-independent
   properties: rerun identity, permutation invariance, ranked rows, phased rows adding up to the MUUG, and the table
   kernels' two grouping paths agreeing."""
import os

import numpy as np
import pytest

import harness
import wmda_scale

pytestmark = pytest.mark.gpu


def test_small_variant_vs_oracle():
    n_haps = 20000
    wmda_scale.ensure(n_haps)
    name = wmda_scale.name_of(n_haps)
    conf = wmda_scale.conf()
    lines = wmda_scale.subjects(6, seed=3, n_haps=n_haps)
    got, glog, imp = harness.run_product(name, conf, lines, tag="w5s", quiet=True)
    exp, elog = harness.run_oracle(name, conf, lines, tag="w5s_orc")
    for k in exp:
        assert got[k] == exp[k], k
    # and a mixed bag on the same multi-population graph
    import synth
    mixed = synth.SubjectGen(wmda_scale.union_rows(n_haps), 9, pops=wmda_scale.POPS).mixed(150)
    got, glog, imp = harness.run_product(name, conf, mixed, tag="w5m", quiet=True)
    exp, elog = harness.run_oracle(name, conf, mixed, tag="w5m_orc")
    for k in exp:
        assert got[k] == exp[k], k


def test_full_size_graph_vs_reference_golden():
    """The REAL reference's answer on the full-size graph (1.06 M nodes, three populations): tests/golden/wmda_full holds the
    six output files and per-subject counts it produced in the build container for six high-ambiguity subjects (8 / 16
    alternatives per locus and side; tools/make_golden_wmda.py, 6 minutes of reference time).  Byte for byte."""
    import json

    gdir = os.path.join(harness.GOLD, "wmda_full")
    meta = json.load(open(os.path.join(gdir, "meta.json")))
    assert meta["n_haps"] == wmda_scale.N_HAPS
    wmda_scale.ensure()
    name = wmda_scale.name_of()
    lines = [l.rstrip("\n") for l in open(os.path.join(gdir, "input.csv"))]
    assert lines == wmda_scale.subjects(len(lines))  # the generator still makes the subjects the fixture was made from
    got, glog, imp = harness.run_product(name, wmda_scale.conf(), lines, tag="w5g", quiet=False)
    assert not imp.unsupported
    for k, f in harness.OUT_FILES.items():
        p = os.path.join(gdir, f)
        exp = open(p).read() if os.path.exists(p) else ""
        assert got[k] == exp, k
    elog = [l for l in open(os.path.join(gdir, "log.txt")).read().splitlines() if "Subject:" in l]
    assert glog == elog


def test_full_size_graph_properties(monkeypatch):
    work = wmda_scale.ensure()
    name = wmda_scale.name_of()
    conf = wmda_scale.conf()
    lines = wmda_scale.subjects(48, seed=5)
    got, glog, imp = harness.run_product(name, conf, lines, tag="w5", quiet=True)
    a = imp.netGraph.arrays
    assert a["n_nodes"] >= 1_000_000 and len(a["a_nbr"]) >= 6_000_000 and a["n_pops"] == 3
    assert not imp.unsupported and got["problem"] == ""
    umug = [l.split(",") for l in got["umug"].splitlines()]
    ids = [l.split(",")[0] for l in lines]
    seen = [u[0] for u in umug if u[3] == "0"]
    assert seen == [i for i in ids if i in set(seen)]                      # input order, one rank-0 row per imputed subject
    assert len(seen) + len(got["miss"].splitlines()) == len(lines)
    by = {}
    for u in umug:
        by.setdefault(u[0], []).append(float(u[2]))
    assert all(v == sorted(v, reverse=True) for v in by.values())          # ranked
    pm = {}
    for l in got["pmug"].splitlines():
        f = l.split(",")
        pm.setdefault(f[0], []).append(float(f[2]))
    assert all(v == sorted(v, reverse=True) for v in pm.values())
    # rerun identity and permutation invariance
    got2, _, _ = harness.run_product(name, conf, lines, tag="w5b", quiet=True)
    for k in got:
        assert got2[k] == got[k], k
    perm = np.random.default_rng(2).permutation(len(lines))
    got3, _, _ = harness.run_product(name, conf, [lines[i] for i in perm], tag="w5p", quiet=True)
    for k in ("umug", "umug_pops", "pmug", "pmug_pops"):
        assert sorted(got3[k].splitlines()) == sorted(got[k].splitlines()), k
    # the table kernels' bucket path against their hash-table + radix-sort path in HBM scratch
    monkeypatch.setenv("GRIM_TABLES_HBM", "1")
    got4, _, _ = harness.run_product(name, conf, lines, tag="w5h", quiet=True)
    for k in got:
        assert got4[k] == got[k], k


def test_row_pool_grows_instead_of_running_every_chunk_three_times(monkeypatch):
    """Subjects that fill their tables (~40 rows each) against the stream's opening row pool of 32 rows per line: the first
    chunks overflow, the pool doubles (for the slot at once, for the other slots at their next load) and the later chunks
    run ONCE -- not whole + two halves each, which is what a 2 048-line config-5 stream did until round 3 (77 ms per step
    against 34 ms of kernels).  Same bytes as one big chunk whose pool never overflows."""
    n_haps = 20000
    wmda_scale.ensure(n_haps)
    name = wmda_scale.name_of(n_haps)
    conf = wmda_scale.conf()
    lines = wmda_scale.subjects(3300, seed=11, n_haps=n_haps)
    whole, _, imp0 = harness.run_product(name, conf, lines, tag="w5r0", quiet=True)
    assert imp0.last_stats["reruns"] <= 1  # (the pair pool of the table kernels may have to grow once: counted too)
    monkeypatch.setenv("GRIM_CHUNK_LINES", "512")
    monkeypatch.setenv("GRIM_STREAM_DEPTH", "2")
    got, _, imp = harness.run_product(name, conf, lines, tag="w5r1", quiet=True)
    for k in whole:
        assert got[k] == whole[k], k
    st = imp.last_stats
    assert st["chunks"] == 7
    # each of the two slots may grow its row pool once and its pair pool once; without the growth every chunk overflows (>= 7)
    assert 1 <= st["reruns"] <= 4, st
