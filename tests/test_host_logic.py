"""Host-side logic of the product (no GPU): graph arrays, prior matrices, tokeniser, parameters, C-ABI exports."""
import json
import os
import re

import numpy as np
import pytest

import harness


def _cfg(name, **over):
    from grim.run_impute_def import load_config

    work = harness.ensure_graph(name)
    conf = harness.base_conf(harness.POPS[name])
    conf.update(over)
    path = os.path.join(work, "conf_hostlogic.json")
    json.dump(conf, open(path, "w"))
    cwd = os.getcwd()
    os.chdir(work)
    try:
        cfg, _ = load_config(path)
        for k in ("node_file", "top_links_file", "edges_file", "pops_count_file"):
            cfg[k] = os.path.join(work, cfg[k])
    finally:
        os.chdir(cwd)
    return cfg, conf


@pytest.fixture(scope="module")
def graphs():
    import grim_oracle as go
    from grim.imputation.networkx_graph import Graph

    out = {}
    for name in ("cau", "pop4"):
        cfg, _ = _cfg(name)
        g = Graph(cfg).build_graph(cfg["node_file"], cfg["top_links_file"], cfg["edges_file"])
        og = go.OGraph(cfg["full_loci"]).load(cfg["node_file"], cfg["top_links_file"], cfg["edges_file"])
        out[name] = (cfg, g, og)
    return out


@pytest.mark.parametrize("name", ["cau", "pop4"])
def test_csr_arrays_equal_reference_construction(graphs, name):
    cfg, g, og = graphs[name]
    a = g.arrays
    V = a["n_nodes"]
    assert np.array_equal(a["a_start"], og.nbr_start) and np.array_equal(a["a_nbr"], og.nbr)
    assert np.array_equal(a["b_start"], og.w_start[V:]) and np.array_equal(a["b_nbr"], og.w_nbr)
    # the sentinel quirk: the last vertex has no neighbours
    assert a["a_start"][V] == V and a["a_start"][V - 1] > V
    # names round-trip through the 64-bit keys
    for i in range(0, V, 53):
        assert g.key_to_name(a["node_key"][i]) == og.names[i]
        assert np.array_equal(a["freq"][i], np.array(og.attr[og.names[i]][1]))
    # connectors
    rng = np.random.default_rng(0)
    for c in rng.integers(0, V, 500):
        nm = og.names[c]
        lab = og.attr[nm][0]
        for s, ch in enumerate(cfg["full_loci"]):
            if ch in lab:
                continue
            conn = "".join(sorted(lab + ch)) + nm
            code = a["b_conn"][c * 5 + s]
            if conn in og.w_attr:
                assert og.w_attr[conn] == V + code
            else:
                assert code == 0xFFFFFFFF
    # label index
    for m in range(32):
        lab = "".join(cfg["full_loci"][s] for s in range(5) if (m >> s) & 1)
        ids = a["lab_nodes"][a["lab_start"][m]: a["lab_start"][m + 1]]
        assert [og.names[i] for i in ids] == og.haps_by_label(lab)


def test_prior_matrices_bit_identical_to_oracle(graphs):
    import grim_oracle as go
    from grim.imputation.impute import Imputation

    cfg, g, og = graphs["pop4"]
    ocfg = go.config_from_json(harness.base_conf(harness.POPS["pop4"]))
    ocfg["pops_count_file"] = cfg["pops_count_file"]
    for unk in ("MR", "SR"):
        cfg2 = dict(cfg, UNK_priors=unk)
        ocfg2 = dict(ocfg, UNK_priors=unk)
        imp = Imputation(g, cfg2)
        oimp = go.OracleImputer(og, ocfg2)
        cases = [("CAU", "AFA"), ("CAU", "CAU"), ("UNK", "HIS"), ("API", "UNK"), ("UNK", "UNK"), ("", ""),
                 ("CAU;AFA", "HIS;API"), ("CAU;XXX", "API"), ("AFA;AFA", "AFA"), ("HIS", "CAU;HIS;API")]
        for r1, r2 in cases:
            oimp.impute_one("", r1, r2)  # empty GL: only the prior is computed
            m = imp._prior_matrix(r1, r2, cfg["priority"])
            assert m.tobytes() == np.ascontiguousarray(oimp.prior, dtype=np.float64).tobytes(), (unk, r1, r2)


def test_tokeniser(graphs):
    from grim.imputation import impute as I

    cfg, g, og = graphs["cau"]
    imp = I.Imputation(g, cfg)
    assert I.clean_up_gl("A*01:01g+A*02:01L^B*UUUU+B*UUUU^C*07:01+C*07:02") == "A*01:01+A*02:01^C*07:01+C*07:02"
    kind, (n, slots, same, pos) = imp._tokenise("B*08:01+B*07:02^A*01:01/A*01:01/A*02:01+A*02:01", True)
    assert kind == I._DEV and n == 2 and slots == [0, 1] and same == 0
    (ids0, w0), (ids1, w1) = pos[0]
    assert w0 == 3 and len(ids0) == 2 and w1 == 1  # duplicates collapse, the original width is kept
    assert imp._tokenise("A*01:01^B*08:01+B*07:02", True)[0] == I._PROBLEM_ID  # no '+'
    assert imp._tokenise("", True)[0] == I._PROBLEM_ID
    kind, (n, slots, same, pos) = imp._tokenise("A*01:01+A*01:01^B*08:01+B*07:02", True)
    assert same == 1
    with pytest.raises(KeyError):
        imp._tokenise("X*01:01+X*01:02", True)
    assert imp._tokenise("X*01:01+X*01:02", False)[0] == I._MISS_NO_DEVICE
    with pytest.raises(Exception):
        imp._tokenise("A*01:01+A*02:01^^B*08:01+B*07:02", True)  # empty entry -> IndexError like the reference
    # an allele the graph does not know gets a fresh id past the graph's own
    kind, (n, slots, same, pos) = imp._tokenise("A*99:99+A*02:01", True)
    assert pos[0][0][0][0] >= g.n_graph_alleles[0]


def test_ladder_and_params(graphs):
    from grim.imputation.impute import Imputation

    cfg, g, og = graphs["cau"]
    imp = Imputation(g, cfg)
    p = imp._params(cfg, True, False)
    lad = [p.ladder[i] for i in range(p.n_ladder)]
    e, ref = cfg["epsilon"], []
    while e > 0:
        e /= 10
        if e < 1.0e-9:
            e = 0.0
        ref.append(e)
    assert lad == ref and lad[-1] == 0.0
    assert p.planb_rows == 6 and p.planb_nblk[1] == 2 and p.planb_blk[1][0] == 0b00111 and p.planb_blk[1][1] == 0b11000


def test_c_abi_library_exports_every_declared_symbol():
    import __graft_entry__ as ge
    from grim import _native

    ge.build()
    header = open(os.path.join(harness.ROOT, "include", "grim_hip.h")).read()
    declared = set(re.findall(r"\b(grim_[a-z_]+)\s*\(", header))
    assert declared == set(_native.EXPORTS)
    L = _native.lib()
    for name in declared:
        assert getattr(L, name) is not None
    assert _native.SUBJECT_DT.itemsize == 64 and _native.RESULT_DT.itemsize == 56 and _native.ROW_DT.itemsize == 32


def test_product_fails_loudly_without_gpu(graphs):
    """No CPU fallback: on a machine without a HIP device the product path raises."""
    from grim import _native
    from grim.imputation.impute import Imputation

    if _native.lib().grim_device_count() > 0:
        pytest.skip("a GPU is present")
    cfg, g, og = graphs["cau"]
    imp = Imputation(g, cfg)
    with pytest.raises(_native.NativeError):
        imp.impute_one("S", "A*01:01+A*02:01", [1, 1, 1, 1], "CAU", "CAU", cfg["priority"], 1e-3, 1000, True, True, True, False)


def test_open_gl_string_hook_equals_reference_vectors():
    """Imputation.open_gl_string (the EM package's hook, impute.py:305-351) against vectors taken from the
    reference (tests/golden/open_gl_string.json; made by calling the reference's method on these inputs)."""
    from grim.imputation.impute import Imputation

    imp = Imputation.__new__(Imputation)
    for case in json.load(open(os.path.join(harness.GOLD, "open_gl_string.json"))):
        try:
            got = ["ok", imp.open_gl_string(case["gl"], case["cutoff"])]
        except Exception as e:  # the reference raises IndexError on an empty locus entry
            got = ["exc", type(e).__name__]
        assert got == case["result"], case["gl"]
