"""The CPU oracle against every golden vector produced by the real reference (tools/make_golden.py)."""
import pytest

import harness


@pytest.mark.parametrize("scenario", harness.scenarios())
def test_oracle_matches_reference_outputs(scenario):
    gname, conf, lines, exp, elog, em = harness.golden(scenario)
    got, glog = harness.run_oracle(gname, conf, lines, tag="t_orc_" + scenario, em_mr=em)
    for k in exp:
        assert got[k] == exp[k], "%s: %s differs from the reference output" % (scenario, k)
    assert glog == elog, "%s: per-subject haplotype counts differ" % scenario


def test_known_answer_readme():
    """README.md:123-124 of the reference: D1 -> 8400 phased pairs, 6028 MUUGs."""
    gname, conf, lines, exp, elog, em = harness.golden("cau_min")
    assert elog == ["0 Subject: D1 8400 haplotypes", "0 Subject: D1 6028 haplotypes"]
    assert exp["umug"].splitlines()[0] == (
        "D1,A*01:02+A*02:01^B*15:01+B*15:01^C*03:03+C*03:04^DQB1*03:02+DQB1*06:02^DRB1*04:01+DRB1*15:01,"
        "8.838563003520004e-17,0")


@pytest.mark.parametrize("name", ["cau", "pop4"])
def test_oracle_intermediates_equal_reference_dump(name, monkeypatch):
    """tests/golden/intermediates/<name>.json (tools/make_golden_intermediates.py: recording wrappers around the REAL
    reference's gen_phases / open_phases / convert_list_to_one_dim): the oracle's phases, candidate lists IN ORDER and top
    lists, call by call, for ~30 subjects per graph -- so that a change which moves an intermediate fails here, at the
    intermediate, and not only at an output file."""
    import hashlib
    import json
    import os

    import grim_oracle as go

    fx = json.load(open(os.path.join(harness.GOLD, "intermediates", name + ".json")))

    def digest(x, full=12):
        if len(x) <= full:
            return x
        return {"n": len(x), "sha256": hashlib.sha256(repr(x).encode()).hexdigest(), "head": x[:2], "tail": x[-1:]}

    work = harness.ensure_graph(fx["graph"])
    conf, cpath = harness._write_inputs(work, fx["conf"], fx["lines"], "t_int_" + name)
    cwd = os.getcwd()
    os.chdir(work)
    try:
        cfg = go.config_from_json(conf)
        g = go.OGraph(cfg["full_loci"]).load(cfg["node_file"], cfg["top_links_file"], cfg["edges_file"])
        imp = go.OracleImputer(g, cfg)
        per, state = {}, {"sid": None}

        def cur():
            return per.setdefault(state["sid"], {"phases": None, "open": [], "top": []})

        orig_phases, orig_open, orig_top, orig_one = go.phases_of, go.OracleImputer._open, go.OracleImputer._top, go.OracleImputer.impute_one

        def phases_of(gen, n_loci, b_phases=None):
            out = orig_phases(gen, n_loci, b_phases)
            if cur()["phases"] is None:
                cur()["phases"] = [[list(h1), list(h2)] for h1, h2 in out]
            return out

        def _open(self, pmags, n_loci):
            out = orig_open(self, pmags, n_loci)
            cur()["open"].append([[digest([list(c) for c in side[0]]) for side in ph[:2]] for ph in out])
            return out

        def _top(self, probs):
            out = orig_top(self, probs)
            cur()["top"].append(digest([[p, [int(k), int(j)]] for p, k, j in out], 2))
            return out

        monkeypatch.setattr(go, "phases_of", phases_of)
        monkeypatch.setattr(go.OracleImputer, "_open", _open)
        monkeypatch.setattr(go.OracleImputer, "_top", _top)
        for line in fx["lines"]:
            parts = line.split(",")
            state["sid"] = parts[0]
            imp.plan = "a"
            try:
                imp.impute_one(parts[1], parts[2] if len(parts) > 2 else None, parts[3] if len(parts) > 3 else None,
                               b_phases=[1] * (len(imp.full_loci) - 1))
            except Exception:
                pass  # (the reference's bare except: the line goes to .problem; what was recorded until then still counts)
    finally:
        os.chdir(cwd)
    exp = fx["subjects"]
    assert sorted(per) == sorted(exp)
    for sid in exp:
        assert per[sid]["phases"] == exp[sid]["phases"], (sid, "gen_phases")
        assert per[sid]["open"] == exp[sid]["open"], (sid, "open_phases")
        tops = per[sid]["top"]
        assert len(tops) == exp[sid]["top"]["calls"], (sid, "number of convert_list_to_one_dim calls")
        assert tops[:4] == exp[sid]["top"]["first"], (sid, "first top lists")
        assert hashlib.sha256(repr(tops).encode()).hexdigest() == exp[sid]["top"]["sha256"], (sid, "top lists")


def test_every_golden_directory_is_named_in_a_committed_generator():
    """tests/golden/<dir> is data a committed script made from the real reference: every scenario directory must be the
    target of a `run_scenario("<dir>", ...)` call in tools/make_golden*.py, every other entry of tests/golden/ must be
    named there too -- so `python tools/make_golden.py` (or GOLDEN_ONLY=<dir>) can always re-make the committed bytes."""
    import glob
    import os
    import re

    src = "".join(open(f).read() for f in sorted(glob.glob(os.path.join(harness.HERE, "make_golden*.py"))))
    named = set(re.findall(r'run_scenario\(\s*"([A-Za-z0-9_]+)"', src))
    missing = [s for s in harness.scenarios() if s not in named]
    assert not missing, "golden scenarios without a generator call: %s" % missing
    for entry in sorted(os.listdir(harness.GOLD)):
        if entry in named:
            continue
        assert entry.split(".")[0] in src, "tests/golden/%s is not named in any tools/make_golden*.py" % entry
