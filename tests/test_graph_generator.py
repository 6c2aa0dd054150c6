"""produce_hpf / generate_graph (product, drop-in for graph_generation/) against the reference's CSV md5s."""
import hashlib
import json
import os

import pytest

import harness


def _md5(path, sort_lines=False):
    data = open(path, "rb").read()
    if sort_lines:
        lines = data.split(b"\n")
        data = b"\n".join([lines[0]] + sorted(lines[1:]))
    return hashlib.md5(data).hexdigest()


@pytest.mark.parametrize("name", ["cau", "pop4", "cau_bc", "pop4_bc"])
def test_generated_csv_identical_to_reference(name):
    work = harness.ensure_graph(name)
    info = json.load(open(os.path.join(harness.GOLD, "graphs", name, "graph_info.json")))
    got = {
        "hpf.csv": _md5(os.path.join(work, "output", "hpf.csv")),
        "pop_counts_file.txt": _md5(os.path.join(work, "output", "pop_counts_file.txt")),
        "nodes.csv": _md5(os.path.join(work, "output", "csv", "nodes.csv")),
        "edges.csv": _md5(os.path.join(work, "output", "csv", "edges.csv")),
        "top_links.csv(sorted rows)": _md5(os.path.join(work, "output", "csv", "top_links.csv"), True),
        "info_node.csv": _md5(os.path.join(work, "output", "csv", "info_node.csv")),
    }
    assert got == info["md5"]
    assert open(os.path.join(work, "output", "pop_counts_file.txt")).read() == open(
        os.path.join(harness.GOLD, "graphs", name, "pop_counts_file.txt")).read()


def _gen_both(work, conf, tmp_path):
    """generate_graph through the library's C++ generator and through the Python twin -> two dirs of CSVs"""
    from graph_generation.generate_neo4j_multi_hpf import generate_graph

    out = {}
    cwd = os.getcwd()
    os.chdir(work)
    try:
        for tag, twin in (("cpp", False), ("py", True)):
            c = dict(conf, graph_files_path=str(tmp_path / tag) + "/")
            cpath = str(tmp_path / (tag + ".json"))
            json.dump(c, open(cpath, "w"))
            generate_graph(cpath, quiet=True, python_twin=twin)
            out[tag] = str(tmp_path / tag)
    finally:
        os.chdir(cwd)
    return out


def _load_both(conf_path, csvdir):
    import numpy as np
    from grim.imputation.networkx_graph import Graph
    from grim.run_impute_def import load_config

    cfg, _ = load_config(conf_path)
    files = [os.path.join(csvdir, f) for f in ("nodes.csv", "top_links.csv", "edges.csv")]
    a = Graph(cfg).build_graph(*files).arrays
    b = Graph(cfg)._build_graph_python(*files).arrays
    assert set(a) == set(b)
    for k in b:
        if isinstance(b[k], np.ndarray):
            assert a[k].dtype == b[k].dtype and np.array_equal(a[k], b[k]), k
        else:
            assert a[k] == b[k], k


@pytest.mark.parametrize("name", ["cau", "pop4"])
def test_cpp_generator_and_loader_equal_python_twins(name, tmp_path):
    work = harness.ensure_graph(name)
    conf = json.load(open(os.path.join(work, "graph_conf.json")))
    dirs = _gen_both(work, conf, tmp_path)
    info = json.load(open(os.path.join(harness.GOLD, "graphs", name, "graph_info.json")))["md5"]
    for f in ("nodes.csv", "edges.csv", "top_links.csv", "info_node.csv"):
        assert open(os.path.join(dirs["cpp"], f), "rb").read() == open(os.path.join(dirs["py"], f), "rb").read(), f
    for f in ("nodes.csv", "edges.csv", "info_node.csv"):  # and both equal the reference's own files
        assert _md5(os.path.join(dirs["cpp"], f)) == info[f]
    assert _md5(os.path.join(dirs["cpp"], "top_links.csv"), True) == info["top_links.csv(sorted rows)"]
    _load_both(os.path.join(work, "graph_conf.json"), dirs["cpp"])


def test_cpp_generator_odd_rows(tmp_path):
    """rows below the cutoff, zero frequencies, a repeated (haplotype, population) row, 'g' suffixes, a population
    a haplotype never appears in (the integer 0 of the reference's running sums) -- C++ and Python twin agree"""
    work = tmp_path / "w"
    (work / "output").mkdir(parents=True)
    rows = [
        "hap,pop,freq",
        "A*01:01g~B*08:01g~C*07:01g~DQB1*02:01~DRB1*03:01,CAU,0.05",
        "A*01:01g~B*08:01g~C*07:01g~DQB1*02:01~DRB1*03:01,AFA,0.0125",
        "A*02:01~B*07:02~C*07:02~DQB1*06:02~DRB1*15:01,CAU,0.03",
        "A*02:01~B*07:02~C*07:02~DQB1*06:02~DRB1*15:01,CAU,0.031",      # repeated: the later value wins
        "A*03:01~B*07:02~C*07:02~DQB1*06:02~DRB1*15:01,AFA,1e-09",       # below the cutoff
        "A*03:01~B*35:01~C*04:01~DQB1*03:01~DRB1*01:01,AFA,0.0",         # zero
        "A*02:01~B*44:02~C*05:01~DQB1*03:01~DRB1*04:01,AFA,7.25e-05",
        "A*02:01~B*44:02~C*05:01~DQB1*03:01~DRB1*04:01,HIS,0.00019999999999999998",
        "A*24:02~B*07:02~C*07:02~DQB1*06:02~DRB1*15:01,HIS,3.3333333333333335e-05",
    ]
    (work / "output" / "hpf.csv").write_text("\n".join(rows) + "\n")
    (work / "output" / "pop_counts_file.txt").write_text("CAU,100,0.5\nAFA,50,0.25\nHIS,50,0.25\n")
    conf = harness.base_conf(["CAU", "AFA", "HIS"])
    dirs = _gen_both(str(work), conf, tmp_path)
    for f in ("nodes.csv", "edges.csv", "top_links.csv", "info_node.csv"):
        assert open(os.path.join(dirs["cpp"], f), "rb").read() == open(os.path.join(dirs["py"], f), "rb").read(), f
    nodes = open(os.path.join(dirs["cpp"], "nodes.csv")).read()
    assert ";0;" in nodes or ",0;" in nodes  # an untouched integer 0 is printed as "0", not "0.0"
    cpath = str(tmp_path / "cpp.json")
    cwd = os.getcwd()
    os.chdir(str(work))
    try:
        _load_both(cpath, dirs["cpp"])
    finally:
        os.chdir(cwd)


def test_cpp_generator_reports_errors(tmp_path):
    from grim import _native as nat

    with pytest.raises(ValueError):
        nat.graphgen_csv(str(tmp_path / "missing.csv"), ["CAU"], [1e-5], {"A": 1, "B": 2}, *[str(tmp_path / f) for f in "abcd"])
    bad = tmp_path / "hpf.csv"
    bad.write_text("hap,pop,freq\nA*01:01~B*08:01,XXX,0.1\n")
    with pytest.raises(ValueError):  # the reference dies with KeyError on a population it was not configured for
        nat.graphgen_csv(str(bad), ["CAU"], [1e-5], {"A": 1, "B": 2}, *[str(tmp_path / f) for f in "abcd"])


import pytest as _pytest


@_pytest.mark.parametrize("gname", ["cau", "pop4"])
def test_loader_arrays_equal_reference_dump(gname):
    """The C++ loader's arrays (grim_hostgraph_load_csv) against what the REFERENCE's Graph.build_graph holds for the same
    CSVs (tests/golden/graphs/<name>/loader_arrays.json, written by tools/make_golden.py from the real reference): vertex
    order, plan-A CSR with its sentinel quirk, plan-B CSR, connector pseudo-vertices in creation order."""
    import hashlib
    import json
    import os

    import numpy as np

    import harness
    from grim import _native as nat
    from grim.imputation.networkx_graph import Graph
    from grim.run_impute_def import load_config

    ref = json.load(open(os.path.join(harness.GOLD, "graphs", gname, "loader_arrays.json")))
    work = harness.ensure_graph(gname)
    conf2, cpath = harness._write_inputs(work, harness.base_conf(harness.POPS[gname]), [], "la")
    cwd = os.getcwd()
    os.chdir(work)
    try:
        cfg, _ = load_config(cpath)
        g = Graph(cfg).build_graph(cfg["node_file"], cfg["top_links_file"], cfg["edges_file"])
    finally:
        os.chdir(cwd)
    A = g.arrays
    V = A["n_nodes"]

    def check_arr(a, want):
        a = np.ascontiguousarray(a, dtype="<u4")
        assert int(a.size) == want["n"]
        assert [int(x) for x in a[:8]] == want["head"] and [int(x) for x in a[-8:]] == want["tail"]
        assert hashlib.sha256(a.tobytes()).hexdigest() == want["sha256"]

    def check_names(v, want):
        assert len(v) == want["n"] and v[:3] == want["head"] and v[-3:] == want["tail"]
        assert hashlib.sha256("\n".join(v).encode()).hexdigest() == want["sha256"]

    assert V == ref["n_vertices"]
    check_names([g.node_name(i) for i in range(V)], ref["Vertices"])
    check_arr(A["a_nbr"], ref["Edges"])
    check_arr(A["a_start"], ref["Neighbors_start"])
    check_arr(A["b_nbr"], ref["Whole_Edges"])
    # the reference's Whole_Neighbors_start covers real vertices and connectors; the device keeps the connector rows (the
    # children reach their connectors through b_conn) -- the real vertices' rows are rebuilt here from b_conn to compare all
    n_conn = len(A["b_start"]) - 1
    assert V + n_conn == ref["n_whole_vertices"]
    conn_names = [None] * n_conn
    b_conn = A["b_conn"].reshape(V, nat.MAXL)
    child_rows = [[] for _ in range(V)]
    for i in range(V):
        for s in range(nat.MAXL):
            c = int(b_conn[i, s])
            if c != 0xFFFFFFFF:
                conn_names[c] = g._mask_label(int(A["node_mask"][i]) | (1 << s)) + g.node_name(i)
                child_rows[i].append(V + c)
    check_names(conn_names, ref["Whole_Vertices_connectors"])
    # Whole_Edges = [edges of real vertices (child -> its connectors, ascending)] + [connector -> parents]: the first part
    # must be exactly the children's connector lists in vertex order
    flat = [x for row in child_rows for x in sorted(row)]
    assert [int(x) for x in A["b_nbr"][:len(flat)]] == flat
    starts = np.asarray(A["b_start"], dtype=np.int64)
    assert int(starts[0]) == len(flat)  # the connector rows follow the children's rows
    # full Whole_Neighbors_start, with the loader's forward fill and sentinel (networkx_graph.py:157-198)
    counts = np.array([len(r) for r in child_rows], dtype=np.int64)
    real_starts = np.concatenate([[0], np.cumsum(counts)[:-1]])
    real_starts = np.where(counts > 0, real_starts, -1)
    idx = np.maximum.accumulate(np.where(real_starts >= 0, np.arange(V), -1))
    filled = np.where(idx >= 0, real_starts[np.maximum(idx, 0)], 0)
    whole = np.concatenate([filled, starts[:-1], [V + n_conn]]).astype("<u4")
    check_arr(whole, ref["Whole_Neighbors_start"])


@pytest.mark.parametrize("name", ["cau", "pop4"])
def test_hpf_to_graph_without_the_csv_files(name, tmp_path, monkeypatch):
    """grim.graph_from_freqs (grim_hostgraph_from_hpf: generator and loader back to back in memory) builds the arrays
    graph_freqs() + Graph.build_graph build through the four files; with write_csv the files are the same bytes too"""
    import numpy as np
    from grim import grim
    from grim.imputation.networkx_graph import Graph
    from grim.run_impute_def import load_config

    work = harness.ensure_graph(name)
    conf = json.load(open(os.path.join(work, "graph_conf.json")))
    conf["graph_files_path"] = str(tmp_path / "csv") + "/"
    cpath = str(tmp_path / "direct.json")
    json.dump(conf, open(cpath, "w"))
    monkeypatch.chdir(work)
    cfg, _ = load_config(os.path.join(work, "graph_conf.json"))
    ref = Graph(cfg).build_graph(cfg["node_file"], cfg["top_links_file"], cfg["edges_file"])
    direct = grim.graph_from_freqs(cpath)
    assert not os.path.exists(tmp_path / "csv")  # nothing written
    assert set(direct.arrays) == set(ref.arrays)
    for k, v in ref.arrays.items():
        if isinstance(v, np.ndarray):
            assert direct.arrays[k].dtype == v.dtype and np.array_equal(direct.arrays[k], v), k
        else:
            assert direct.arrays[k] == v, k
    assert direct.n_graph_alleles == ref.n_graph_alleles
    for s in range(len(ref.full_loci)):  # same allele numbering: a subject tokenizes to the same ids on either graph
        assert [ref.adict.name(s, i) for i in range(ref.n_graph_alleles[s])] == \
               [direct.adict.name(s, i) for i in range(direct.n_graph_alleles[s])]
    grim.graph_from_freqs(cpath, write_csv=True)
    for f in ("nodes.csv", "edges.csv", "top_links.csv", "info_node.csv"):
        assert open(tmp_path / "csv" / f, "rb").read() == open(os.path.join(work, "output", "csv", f), "rb").read(), f
