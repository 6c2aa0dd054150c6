"""Host-side pieces of the drop-in surface that need no GPU: the Graph look-ups of networkx_graph.py:215-321 over the loaded
arrays and the module-level writers of impute.py:24-99, each against the oracle's restatement (which the reference's golden
outputs pin)."""
import io
import os

import numpy as np
import pytest

import harness


def _graphs(gname):
    import grim_oracle as go
    from grim.imputation.networkx_graph import Graph
    from grim.run_impute_def import load_config

    work = harness.ensure_graph(gname)
    conf = harness.base_conf(harness.POPS[gname])
    conf2, cpath = harness._write_inputs(work, conf, [], "bq")
    cwd = os.getcwd()
    os.chdir(work)
    try:
        cfg, _ = load_config(cpath)
        g = Graph(cfg).build_graph(cfg["node_file"], cfg["top_links_file"], cfg["edges_file"])
        og = go.OGraph(cfg["full_loci"]).load(cfg["node_file"], cfg["top_links_file"], cfg["edges_file"])
    finally:
        os.chdir(cwd)
    return g, og


@pytest.mark.parametrize("gname", ["cau", "pop4"])
def test_graph_lookups_equal_oracle(gname):
    g, og = _graphs(gname)
    rng = np.random.default_rng(3)
    names = list(og.attr.keys())
    sample = [names[int(i)] for i in rng.choice(len(names), size=400, replace=False)]
    sample += ["A*99:99", "A*01:01~B*99:99", "B*08:01~A*01:01", "", "Q*01:01", names[0] + "~X"]
    # adjs_query: keys in the same (insertion) order, same frequencies
    a, b = g.adjs_query(sample), og.adjs_query(sample)
    assert list(a.keys()) == list(b.keys())
    assert all(a[k] == [float(x) for x in b[k]] for k in a)
    # labels
    for label in ("1", "12", "125", "2345", "12345", "45", "9", ""):
        assert g.haps_by_label(label) == og.haps_by_label(label), label
        hp = g.haps_with_probs_by_label(label)
        assert list(hp.keys()) == og.haps_by_label(label)
        assert all(hp[k] == [float(x) for x in og.attr[k][1]] for k in hp)
    # plan-B look-ups: every (child, parent label) combination of the sampled nodes
    full = og.full_loci
    for nm in sample[:150]:
        if nm not in og.attr:
            continue
        la = og.attr[nm][0]
        for lb in [la] + ["".join(sorted(la + c)) for c in full if c not in la] + ["12345", "1"]:
            a, b = g.adjs_query_by_color([nm, "A*99:99"], la, lb), og.adjs_query_by_color([nm, "A*99:99"], la, lb)
            assert list(a.keys()) == list(b.keys()), (nm, la, lb)
            assert all(a[k] == [float(x) for x in b[k]] for k in a)
    assert g.node_probs(sample[:50], "x") == {k: [float(x) for x in og.w_attr[k][1]] for k in sample[:50] if k in og.w_attr}


def test_module_writers_equal_oracle():
    import grim_oracle as go
    from grim.imputation import impute as I

    rng = np.random.default_rng(5)
    haps = ["A*01:01~B*08:01", "A*02:01~B*07:02", "A*03:01~B*35:01", "A*24:02~B*44:02"]
    pops = ["CAU", "AFA", "HIS"]
    for trial in range(30):
        n = int(rng.integers(1, 40))
        pairs = [[haps[int(rng.integers(0, 4))], haps[int(rng.integers(0, 4))]] for _ in range(n)]
        races = [[pops[int(rng.integers(0, 3))], pops[int(rng.integers(0, 3))]] for _ in range(n)]
        probs = [float(x) for x in rng.choice([1e-9, 2.5e-9, 3e-10, 7.25e-8], size=n)]  # ties on purpose
        limit = int(rng.integers(1, 12))
        for sign in (",", "+"):
            out, exp = io.StringIO(), []
            I.write_best_prob("S1", pairs, probs, limit, out, sign)
            go._write_merged("S1", pairs, probs, limit, exp, sign)
            assert out.getvalue() == "".join(exp)
        table = {}
        for (a, b), p in zip(pairs, probs):
            table[a + "^" + b] = table.get(a + "^" + b, 0.0) + p
        out, exp = io.StringIO(), []
        I.write_best_prob_genotype("S1", table, limit, out)
        go._write_sorted("S1", table, limit, exp)
        assert out.getvalue() == "".join(exp)
        out, exp = io.StringIO(), []
        I.write_best_hap_race_pairs("S1", pairs, races, probs, limit, out)
        go._write_pairs_with_races("S1", pairs, races, probs, limit, exp)
        assert out.getvalue() == "".join(exp)
