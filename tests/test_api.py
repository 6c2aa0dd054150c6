"""The top-level drop-in API (reference grim/grim.py:40-87, README.md:46-125) on the GPU: produce_hpf -> graph_freqs ->
impute, graph reuse, the packaged-configuration path, the banner; and two batches alive on one context."""
import contextlib
import io
import json
import os
import shutil

import numpy as np
import pytest

import harness
import synth

pytestmark = pytest.mark.gpu


def _texts(d):
    return {k: (open(os.path.join(d, f)).read() if os.path.exists(os.path.join(d, f)) else "") for k, f in harness.OUT_FILES.items()}


def test_readme_flow_graph_freqs_impute_and_graph_reuse(tmp_path, monkeypatch):
    """README.md:46-125 with conf/minimal-configuration.json: the known answer (D1 -> 8400 / 6028, reference outputs
    byte for byte), the banner, then impute(conf2, graph=g) on the returned graph object."""
    from graph_generation.generate_hpf import produce_hpf
    from grim import grim

    monkeypatch.delenv("GRIM_QUIET", raising=False)
    work = tmp_path
    os.makedirs(work / "data" / "freqs")
    os.makedirs(work / "data" / "subjects")
    os.makedirs(work / "conf")
    shutil.copy(os.path.join(harness.GOLD, "data", "freqs", "CAU.freqs.gz"), work / "data" / "freqs")
    shutil.copy(os.path.join(harness.GOLD, "data", "subjects", "donor.csv"), work / "data" / "subjects")
    gname, conf, lines, exp, elog, em = harness.golden("cau_min")
    conf["imputation_in_file"] = "data/subjects/donor.csv"
    json.dump(conf, open(work / "conf" / "minimal-configuration.json", "w"))
    monkeypatch.chdir(work)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        produce_hpf(conf_file="conf/minimal-configuration.json")
        grim.graph_freqs(conf_file="conf/minimal-configuration.json")
        g = grim.impute(conf_file="conf/minimal-configuration.json")
    out = buf.getvalue()
    for frag in ("Conversion to HPF file based on following configuration:", "Performing graph generation based on following configuration:",
                 "Performing imputation based on:", "\tPopulation: ['CAU']", "\tUNK priority: SR", "\tEpsilon: 0.001",
                 "\tNodes File: output/csv/nodes.csv", "\tTop Links File: output/csv/edges.csv",
                 "\tLoci Map: {'A': 1, 'B': 2, 'C': 3, 'DQB1': 4, 'DRB1': 5}", "\tSave space mode: False",
                 "0 Subject: D1 8400 haplotypes", "0 Subject: D1 6028 haplotypes"):
        assert frag in out, frag
    got = _texts(work / "output")
    for k in exp:
        assert got[k] == exp[k], k
    # second call on the same graph object, other input, other output directory
    rows = synth.read_freqs(synth.CAU_FREQS)
    lines2 = synth.SubjectGen(rows, 71).mixed(200)
    conf2 = dict(conf, imputation_in_file="data/subjects/second.csv", imputation_out_path="output2")
    open(work / "data" / "subjects" / "second.csv", "w").write("\n".join(lines2) + "\n")
    json.dump(conf2, open(work / "conf" / "second.json", "w"))
    with contextlib.redirect_stdout(io.StringIO()):
        g2 = grim.impute(conf_file="conf/second.json", graph=g)
    assert g2 is g
    exp2, _ = harness.run_oracle("cau", conf2, lines2, tag="api_orc")
    got2 = _texts(work / "output2")
    for k in exp2:
        assert got2[k] == exp2[k], k


def test_packaged_configuration_path(tmp_path, monkeypatch):
    """conf_file == "": the configuration, frequency data and sample subject packaged with the library (grim/grim.py:40-74):
    graph files under <package>/graph_generation/, input <package>/data/subjects/donor.csv, outputs under ./output"""
    from graph_generation import generate_hpf
    from grim import grim

    pkg = harness.PKG
    monkeypatch.setattr(generate_hpf, "project_dir", pkg + "/graph_generation/")
    # the packaged conf's freq_data_dir is relative to the directory produce_hpf prefixes: give it the packaged data
    os.makedirs(os.path.join(pkg, "graph_generation", "data"), exist_ok=True)
    if not os.path.exists(os.path.join(pkg, "graph_generation", "data", "freqs")):
        shutil.copytree(os.path.join(pkg, "data", "freqs"), os.path.join(pkg, "graph_generation", "data", "freqs"))
    # graph_freqs("") reads <package>/graph_generation/output/hpf.csv but writes the CSVs under the conf's relative
    # graph_files_path (generate_neo4j_multi_hpf.py:224-244), and impute("") looks for them under
    # <package>/graph_generation/: as in the reference, the two meet when the graph is generated from that directory
    monkeypatch.chdir(os.path.join(pkg, "graph_generation"))
    with contextlib.redirect_stdout(io.StringIO()):
        generate_hpf.produce_hpf(os.path.join(pkg, "conf", "minimal-configuration.json"))
        grim.graph_freqs()
    monkeypatch.chdir(tmp_path)
    with contextlib.redirect_stdout(io.StringIO()):
        grim.impute()
    gname, conf, lines, exp, elog, em = harness.golden("cau_min")
    got = _texts(tmp_path / "output")
    for k in exp:
        assert got[k] == exp[k], k
    shutil.rmtree(os.path.join(pkg, "graph_generation", "data"), ignore_errors=True)


def test_two_live_batches_on_one_context():
    """grim_batch_upload of a second, larger batch while the first one is alive, then the first one runs again: the
    per-workgroup scratch is bound when a run starts, never kept by a batch."""
    from grim import _native as nat

    rows = synth.read_freqs(synth.CAU_FREQS)
    conf = harness.base_conf(["CAU"])
    got, glog, imp = harness.run_product("cau", conf, ["S0,A*01:01+A*02:01^B*08:01+B*07:02,CAU,CAU"], tag="tl", quiet=True)
    cfg = imp.config
    g = imp.netGraph
    ctx = nat.default_context(None)
    params = imp._params(cfg, cfg["planb"], False)
    ps, keep = nat.prior_spec(cfg["priority"], imp.unk_priors, imp.count_by_prob)

    def batch(lines):
        parsed = nat.Parsed(g.adict, ("\n".join(lines) + "\n").encode(), cfg["planb"])
        priors = np.stack([nat.prior_matrix(ps, cfg["pops"], r1, r2) for r1, r2 in parsed.races()])
        b = nat.DeviceBatch(ctx, g.device(ctx), params, parsed.subjects(), parsed.tokens(), priors)
        parsed.close()
        return b

    small = batch(synth.SubjectGen(rows, 81).mixed(3, amb=0.6, miss=0.4))       # few scratch slots
    small.run()
    res_a, rows_a = small.results()
    big = batch(synth.SubjectGen(rows, 82).mixed(3000, amb=0.5, miss=0.3))     # many more slots: the context's scratch grows
    big.run()
    small.run()                                                                # must not touch freed memory
    res_b, rows_b = small.results()
    for f in ("status", "plan", "plan_phased", "n_pairs", "n_genotypes", "n_rows", "max_prob"):  # (row offsets depend on the
        assert np.array_equal(res_a[f], res_b[f]), f                                          # order workgroups reach the pool)
    nr = int(res_a["n_rows"].sum())
    assert nr > 0
    for t in range(4):
        for ra, rb in zip(res_a, res_b):
            a = rows_a[ra["row_off"][t]: ra["row_off"][t] + ra["n_rows"][t]]
            b = rows_b[rb["row_off"][t]: rb["row_off"][t] + rb["n_rows"][t]]
            assert a.tobytes() == b.tobytes()
    big.run()
    small.close()
    big.close()


def test_graph_from_freqs_serves_impute(tmp_path, monkeypatch):
    """grim.graph_from_freqs (hpf.csv -> arrays, no graph CSVs) gives impute(conf, graph=g) the graph that graph_freqs +
    Graph.build_graph give it: the reference's sample run comes out byte for byte"""
    from graph_generation.generate_hpf import produce_hpf
    from grim import grim

    work = tmp_path
    os.makedirs(work / "data" / "freqs")
    os.makedirs(work / "data" / "subjects")
    shutil.copy(os.path.join(harness.GOLD, "data", "freqs", "CAU.freqs.gz"), work / "data" / "freqs")
    shutil.copy(os.path.join(harness.GOLD, "data", "subjects", "donor.csv"), work / "data" / "subjects")
    gname, conf, lines, exp, elog, em = harness.golden("cau_min")
    conf["imputation_in_file"] = "data/subjects/donor.csv"
    json.dump(conf, open(work / "conf.json", "w"))
    monkeypatch.chdir(work)
    with contextlib.redirect_stdout(io.StringIO()):
        produce_hpf(conf_file="conf.json")
        g = grim.graph_from_freqs("conf.json")
        assert not os.path.exists(work / "output" / "csv" / "nodes.csv")  # no graph CSV was written
        g2 = grim.impute(conf_file="conf.json", graph=g)
    assert g2 is g
    got = _texts(work / "output")
    for k in exp:
        assert got[k] == exp[k], k
