"""GPU parity: the HIP path (through the C-ABI) against the reference's golden outputs, against the
oracle on seeded inputs, and through size-independent properties at BASELINE.json's full size."""
import os

import numpy as np
import pytest

import harness
import synth

pytestmark = pytest.mark.gpu

# probabilities: bit-exact is the target and what is asserted (string equality of repr());
# BASELINE.json's tolerance is 1e-6 relative, ids and ranking exact.
STRICT = os.environ.get("GRIM_ON_UNSUPPORTED", "raise") == "raise"


def _run(gname, conf, lines, tag, em=False):
    mode = "raise" if STRICT else "skip"
    got, glog, imp = harness.run_product(gname, conf, lines, tag=tag, em_mr=em, on_unsupported=mode)
    return got, glog, imp


# the `*_irregular*` scenarios hold GL strings that name a locus twice or mix loci in one entry.  The reference answers them
# (gl2haps pairs the entries by index after a per-side string sort, impute.py:246-272 -- what is in the golden files); this
# build REPORTS exactly these subjects (reason 8) and answers every other line of the file like the reference.
IRREGULAR_IDS = ["I0", "I1", "I2", "I3", "I6", "I7", "I8", "I9"]  # I4 and I5 sort back into regular subjects


@pytest.mark.parametrize("scenario", harness.scenarios())
def test_golden_scenarios(scenario):
    gname, conf, lines, exp, elog, em = harness.golden(scenario)
    if "_irregular" in scenario:
        from grim.imputation.impute import UnsupportedSubjects

        with pytest.raises(UnsupportedSubjects) as ei:
            harness.run_product(gname, conf, lines, tag="t_" + scenario, em_mr=em, on_unsupported="raise")
        assert [(sid, r) for _, sid, r in ei.value.items] == [(sid, 8) for sid in IRREGULAR_IDS]
        assert [lines[i].split(",")[0] for i, _, _ in ei.value.items] == IRREGULAR_IDS  # global line numbers
        got, glog, imp = harness.run_product(gname, conf, lines, tag="t_" + scenario, em_mr=em, on_unsupported="skip")
        assert [sid for _, sid, _ in imp.unsupported] == IRREGULAR_IDS
    else:
        got, glog, imp = _run(gname, conf, lines, "t_" + scenario, em)
    skipped = [sid for _, sid, _ in imp.unsupported]
    exp = harness.drop_subjects(exp, skipped)
    for k in exp:
        assert got[k] == exp[k], "%s: %s differs from the reference output" % (scenario, k)
    if not skipped:
        assert glog == elog


def _against_oracle(gname, conf, lines, tag):
    got, glog, imp = _run(gname, conf, lines, tag)
    exp, elog = harness.run_oracle(gname, conf, lines, tag=tag + "_orc")
    skipped = [sid for _, sid, _ in imp.unsupported]
    exp = harness.drop_subjects(exp, skipped)
    for k in exp:
        assert got[k] == exp[k], "%s: %s differs from the oracle" % (tag, k)
    return imp


def test_seeded_full_vs_oracle():
    rows = synth.read_freqs(synth.CAU_FREQS)
    _against_oracle("cau", harness.base_conf(["CAU"]), synth.SubjectGen(rows, 21).full(3000), "r_full")


def test_seeded_mixed_vs_oracle():
    rows = synth.read_freqs(synth.CAU_FREQS)
    _against_oracle("cau", harness.base_conf(["CAU"]), synth.SubjectGen(rows, 22).mixed(500), "r_mixed")


def test_seeded_pop4_vs_oracle():
    rows = synth.read_freqs(synth.CAU_FREQS)
    conf = harness.base_conf(harness.POPS["pop4"])
    conf["UNK_priors"] = "MR"
    _against_oracle("pop4", conf, synth.SubjectGen(rows, 23, pops=harness.POPS["pop4"]).mixed(500), "r_pop4")


def test_high_ambiguity_both_branches_vs_oracle():
    rows = synth.read_freqs(synth.CAU_FREQS)
    lines = synth.SubjectGen(rows, 24).high_ambiguity(6, width=5)
    conf = harness.base_conf(["CAU"])
    _against_oracle("cau", conf, lines, "r_amb_open")           # 5^5 < 1e5: cartesian opening
    conf2 = dict(conf, number_of_options_threshold=500)
    _against_oracle("cau", conf2, lines, "r_amb_filter")        # label scan


def test_config2_full_size_properties():
    """BASELINE config 2: 10k fully typed subjects.  Size-independent properties."""
    rows = synth.read_freqs(synth.CAU_FREQS)
    lines = synth.SubjectGen(rows, 0).full(10000)
    conf = harness.base_conf(["CAU"])
    got, glog, imp = _run("cau", conf, lines, "c2")
    assert got["miss"] == "" and got["problem"] == ""
    umug = [l.split(",") for l in got["umug"].splitlines()]
    pmug = [l.split(",") for l in got["pmug"].splitlines()]
    ids = [l.split(",")[0] for l in lines]
    assert [u[0] for u in umug] == ids                       # exactly one MUUG per subject, input order
    assert all(u[3] == "0" for u in umug)
    by = {}
    for p in pmug:
        by.setdefault(p[0], []).append(float(p[2]))
    for u in umug:
        ps = by[u[0]]
        assert ps == sorted(ps, reverse=True)                # ranked
        if len(ps) < 10:                                     # all phased pairs listed: they add up to the MUUG
            assert abs(sum(ps) - float(u[2])) <= 1e-12 * float(u[2])
    # idempotence and order independence
    perm = np.random.default_rng(5).permutation(len(lines))
    got2, _, _ = _run("cau", conf, [lines[i] for i in perm], "c2p")
    a = sorted(got["umug"].splitlines())
    b = sorted(got2["umug"].splitlines())
    assert a == b
    assert sorted(got["pmug"].splitlines()) == sorted(got2["pmug"].splitlines())
    # a 2000-subject slice against the oracle
    exp, _ = harness.run_oracle("cau", conf, lines[:2000], tag="c2_orc")
    ids2k = {l.split(",")[0] for l in lines[:2000]}
    for k in ("umug", "umug_pops", "pmug", "pmug_pops"):  # all four result files of the slice (miss / problem are empty)
        mine = [l for l in got[k].splitlines() if l.split(",", 1)[0] in ids2k]
        assert mine == exp[k].splitlines(), k
    assert exp["miss"] == "" and exp["problem"] == ""


def test_reference_shaped_impute_one():
    from grim.imputation.impute import Imputation

    work = harness.ensure_graph("cau")
    conf = harness.base_conf(["CAU"])
    got, glog, imp = _run("cau", conf, ["S0,A*01:01+A*02:01^B*08:01+B*07:02^C*07:01+C*07:02^DQB1*02:01+DQB1*06:02^DRB1*03:01+DRB1*15:01,CAU,CAU"], "one")
    cfg = imp.config
    sid, res_m, res_h = imp.impute_one("S0", "A*01:01+A*02:01^B*08:01+B*07:02^C*07:01+C*07:02^DQB1*02:01+DQB1*06:02^DRB1*03:01+DRB1*15:01",
                                       [1, 1, 1, 1], "CAU", "CAU", cfg["priority"], cfg["epsilon"], 1000, True, True, True, False)
    line = got["umug"].splitlines()[0].split(",")
    assert list(res_m["Haps"].keys())[0] == line[1] and str(list(res_m["Haps"].values())[0]) == line[2]
    assert len(res_h["Haps"]) == len(got["pmug"].splitlines())


def test_small_subject_kernel_equals_general_kernel(monkeypatch):
    """The half-wave fast path (grim_small.h) and the general kernel must agree on every subject the
    fast path accepts -- including homozygous loci, missing race columns and Plan-B hand-over."""
    rows = synth.read_freqs(synth.CAU_FREQS)
    gen = synth.SubjectGen(rows, 31)
    lines = gen.full(1500)
    # homozygous variants: copy side 1 onto side 2 at some loci
    rng = np.random.default_rng(9)
    for i in range(300):
        sid, gl, r1, r2 = lines[i].split(",")
        loci = gl.split("^")
        for k in range(5):
            if rng.random() < 0.4:
                a = loci[k].split("+")[0]
                loci[k] = a + "+" + a
        lines[i] = ",".join([sid, "^".join(loci), r1, r2])
    # recombinants (fully typed, no Plan-A hit) and subjects without race columns
    for i in range(300, 400):
        h1, h2 = gen.recombinant(), gen.draw_hap()
        lines[i] = "R%d,%s,CAU,CAU" % (i, gen.gl(h1, h2))
    for i in range(400, 450):
        lines[i] = ",".join(lines[i].split(",")[:2])
    conf = harness.base_conf(["CAU"])
    fast, _, imp_fast = _run("cau", conf, lines, "small_on")
    monkeypatch.setenv("GRIM_NO_SMALL", "1")
    slow, _, imp_slow = _run("cau", conf, lines, "small_off")
    for k in fast:
        assert fast[k] == slow[k], k
    exp, _ = harness.run_oracle("cau", conf, lines[:600], tag="small_orc")
    n = len(exp["pmug"].splitlines())
    assert fast["pmug"].splitlines()[:n] == exp["pmug"].splitlines()


@pytest.mark.parametrize("scenario", ["cau_mixed", "pop4_edge", "cau_planc", "pop4_em_mr", "cau_muug_only"])
def test_cpp_formatter_equals_python_formatter(scenario):
    """impute_lines (C++ tokenizer + formatter in the library) vs impute_lines_python on device results."""
    gname, conf, lines, exp, elog, em = harness.golden(scenario)
    got, glog, imp = _run(gname, conf, lines, "fmt_" + scenario, em)
    cwd = os.getcwd()
    os.chdir(harness.ensure_graph(gname))
    try:
        py = imp.impute_lines_python([l + "\n" for l in lines], imp.config, em_mr=em)
    finally:
        os.chdir(cwd)
    for k in py:
        assert py[k] == got[k], k


def test_one_wave_kernel_equals_general_kernel(monkeypatch):
    """grim_medium.h (one wave per subject, LDS only) against the general kernel on mixed subjects of both graphs."""
    rows = synth.read_freqs(synth.CAU_FREQS)
    for gname, pops, seed in (("cau", ["CAU"], 41), ("pop4", harness.POPS["pop4"], 42)):
        lines = synth.SubjectGen(rows, seed, pops=pops).mixed(1500) + synth.edge_cases(pops[0])
        conf = harness.base_conf(pops)
        conf["UNK_priors"] = "MR"
        monkeypatch.delenv("GRIM_NO_MEDIUM", raising=False)
        fast, _, _ = _run(gname, conf, lines, "med_on")
        monkeypatch.setenv("GRIM_NO_MEDIUM", "1")
        slow, _, _ = _run(gname, conf, lines, "med_off")
        monkeypatch.delenv("GRIM_NO_MEDIUM", raising=False)
        for k in fast:
            assert fast[k] == slow[k], (gname, k)
    # and with a short top list (cut inside a side)
    conf = dict(harness.base_conf(["CAU"]), max_haplotypes_number_in_phase=3)
    lines = synth.SubjectGen(rows, 43).mixed(600, amb=0.4, miss=0.3)
    fast, _, _ = _run("cau", conf, lines, "med_on3")
    monkeypatch.setenv("GRIM_NO_MEDIUM", "1")
    slow, _, _ = _run("cau", conf, lines, "med_off3")
    for k in fast:
        assert fast[k] == slow[k], k


def test_mid_size_kernel_equals_general_kernel(monkeypatch):
    """grim_mid.h (one workgroup per subject, sides / top lists / pair bitmap in LDS, no HBM scratch) against the general kernel
    (GRIM_NO_MID=1) and the oracle: mixed subjects of both graphs with the one-wave kernel on and off (off: every subject is
    the mid-size kernel's), heavy ambiguity (subjects that outgrow its limits are handed on), overlapping '/' lists (the LDS
    dedup table), a short top list (the cut inside a side), phase masks through the golden scenario cau_bin."""
    rows = synth.read_freqs(synth.CAU_FREQS)

    def both(gname, conf, lines, tag, oracle=False):
        monkeypatch.delenv("GRIM_NO_MID", raising=False)
        fast, _, _ = _run(gname, conf, lines, tag + "_on")
        monkeypatch.setenv("GRIM_NO_MID", "1")
        slow, _, _ = _run(gname, conf, lines, tag + "_off")
        monkeypatch.delenv("GRIM_NO_MID", raising=False)
        for k in fast:
            assert fast[k] == slow[k], (tag, k)
        if oracle:
            exp, _ = harness.run_oracle(gname, conf, lines, tag=tag + "_orc")
            for k in exp:
                assert fast[k] == exp[k], (tag, k, "oracle")

    for gname, pops, seed in (("cau", ["CAU"], 241), ("pop4", harness.POPS["pop4"], 242)):
        gen = synth.SubjectGen(rows, seed, pops=pops)
        lines = gen.mixed(1500) + gen.mixed(400, amb=0.8, miss=0.4) + synth.edge_cases(pops[0]) + synth.plan_c_cases(pops[0])
        conf = harness.base_conf(pops)
        conf["UNK_priors"] = "MR"
        both(gname, conf, lines, "mid_" + gname)
        monkeypatch.setenv("GRIM_NO_MEDIUM", "1")
        both(gname, conf, lines[:900], "midnm_" + gname, oracle=gname == "pop4")
        monkeypatch.delenv("GRIM_NO_MEDIUM", raising=False)
    # overlapping lists (side 2 repeats one of side 1's alternatives) and homozygous subjects: the dedup path
    gen = synth.SubjectGen(rows, 243, pops=harness.POPS["pop4"])
    extra = []
    for k, l in enumerate(gen.mixed(300, amb=0.9, miss=0.2, recomb=0.3)):
        f = l.split(",")
        loci = f[1].split("^")
        for i, loc in enumerate(loci):
            a, b = loc.split("+")
            loci[i] = a + "+" + (a.split("/")[0] + "/" + b if k % 2 == 0 else a)
        extra.append(",".join([f[0] + "x", "^".join(loci)] + f[2:]))
    conf = harness.base_conf(harness.POPS["pop4"])
    conf["UNK_priors"] = "MR"
    monkeypatch.setenv("GRIM_NO_MEDIUM", "1")
    both("pop4", conf, extra, "mid_dup", oracle=True)
    # a short top list (cut inside a side), SR priors
    conf3 = dict(harness.base_conf(harness.POPS["pop4"]), max_haplotypes_number_in_phase=3)
    both("pop4", conf3, gen.mixed(500, amb=0.4, miss=0.3), "mid_top3", oracle=True)
    monkeypatch.delenv("GRIM_NO_MEDIUM", raising=False)
    # and the device really took part: the mid-size kernel completed subjects of the mixed set
    got, _, imp = _run("pop4", conf, gen.mixed(800), "mid_count")
    assert imp.last_stats["n"] == 800


def test_side_mask_dedup_equals_table_dedup(monkeypatch):
    """The general and Plan-B kernels' tiled pair passes (>= 8192 scored pairs) dedup from the entities' side masks and
    positions (pair_pass_sidemask, grim_pair.h) instead of a hash table of pairs in the slot; GRIM_NO_SIDEMASK=1 restores the
    table.  Subjects built to need it: few typed loci (the lists saturate), heavy ambiguity, side 2 repeating side 1's
    alternatives (overlapping lists: duplicates across phases and mirrored inside a phase), homozygous subjects; the
    mid-size kernel off so that they all reach the general kernel.  Both ways, the mid-size kernel's own side-mask dedup,
    and the oracle."""
    rows = synth.read_freqs(synth.CAU_FREQS)
    pops = harness.POPS["pop4"]
    gen = synth.SubjectGen(rows, 343, pops=pops)
    lines = []
    for k, l in enumerate(gen.mixed(260, amb=0.9, miss=0.45, recomb=0.2)):
        f = l.split(",")
        loci = f[1].split("^")
        for i, loc in enumerate(loci):
            a, b = loc.split("+")
            loci[i] = a + "+" + (a.split("/")[0] + "/" + b if k % 3 else a)
        lines.append(",".join([f[0] + "y", "^".join(loci)] + f[2:]))
    lines += gen.mixed(120, amb=0.8, miss=0.5, recomb=0.3)
    conf = harness.base_conf(pops)
    conf["UNK_priors"] = "MR"
    out = {}
    for mode in ("mid", "sidemask", "table"):
        monkeypatch.delenv("GRIM_NO_MID", raising=False)
        monkeypatch.delenv("GRIM_NO_SIDEMASK", raising=False)
        if mode != "mid":
            monkeypatch.setenv("GRIM_NO_MID", "1")
        if mode == "table":
            monkeypatch.setenv("GRIM_NO_SIDEMASK", "1")
        out[mode], _, _ = _run("pop4", conf, lines, "smd_" + mode)
    monkeypatch.delenv("GRIM_NO_MID", raising=False)
    monkeypatch.delenv("GRIM_NO_SIDEMASK", raising=False)
    for k in out["table"]:
        assert out["sidemask"][k] == out["table"][k], ("side mask vs table", k)
        assert out["mid"][k] == out["table"][k], ("mid-size kernel vs table", k)
    exp, _ = harness.run_oracle("pop4", conf, lines[:150], tag="smd_orc")
    ids = {l.split(",")[0] for l in lines[:150]}
    for k in ("umug", "umug_pops", "pmug", "pmug_pops"):
        mine = [l for l in out["sidemask"][k].splitlines() if l.split(",", 1)[0] in ids]
        assert mine == exp[k].splitlines(), k


def test_pair_pass_without_dedup_equals_pair_pass_with_dedup(monkeypatch):
    """Subjects whose two '/' lists are disjoint at every differing position skip the first-wins dedup of the pair passes
    (prepare_lists, grim_plan_a.h: no haplotype can belong to two phase sides).  GRIM_NO_NODUP=1 sends every subject through
    the dedup; both ways must give the same files -- on mixed subjects of both graphs (Plan A, B and C), on subjects whose
    lists OVERLAP or are homozygous (the flag must stay off: side 1 and side 2 share haplotypes), and against the oracle."""
    rows = synth.read_freqs(synth.CAU_FREQS)
    for gname, pops, seed in (("cau", ["CAU"], 141), ("pop4", harness.POPS["pop4"], 142)):
        gen = synth.SubjectGen(rows, seed, pops=pops)
        lines = gen.mixed(1200) + gen.mixed(300, amb=0.8, miss=0.4) + synth.edge_cases(pops[0]) + synth.plan_c_cases(pops[0])
        # overlapping lists: side 2 repeats one of side 1's alternatives; and all-homozygous subjects
        extra = []
        for k, l in enumerate(gen.mixed(200, amb=0.9, miss=0.2, recomb=0.3)):
            f = l.split(",")
            loci = f[1].split("^")
            for i, loc in enumerate(loci):
                a, b = loc.split("+")
                if k % 2 == 0:
                    loci[i] = a + "+" + a.split("/")[0] + "/" + b   # the lists of the two sides share an allele
                else:
                    loci[i] = a + "+" + a                          # homozygous everywhere: ONE phase, identical sides
            extra.append(",".join(["X%d" % k, "^".join(loci)] + f[2:]))
        lines += extra
        conf = harness.base_conf(pops)
        conf["UNK_priors"] = "MR"
        monkeypatch.delenv("GRIM_NO_NODUP", raising=False)
        fast, _, _ = _run(gname, conf, lines, "nodup_on")
        monkeypatch.setenv("GRIM_NO_NODUP", "1")
        slow, _, _ = _run(gname, conf, lines, "nodup_off")
        monkeypatch.delenv("GRIM_NO_NODUP", raising=False)
        for k in fast:
            assert fast[k] == slow[k], (gname, k)
        exp, _ = harness.run_oracle(gname, conf, extra, tag="nodup_orc")
        got, _, _ = _run(gname, conf, extra, "nodup_x")
        for k in exp:
            assert got[k] == exp[k], (gname, k)


def test_config4_style_20k_properties_and_oracle_sample():
    """BASELINE config 4 at reduced size: 4-population graph, 20k subjects with missing loci, ambiguity,
    recombinants (Plan B / C exercised), mixed race columns."""
    rows = synth.read_freqs(synth.CAU_FREQS)
    pops = harness.POPS["pop4"]
    lines = synth.SubjectGen(rows, 3, pops=pops).mixed(20000)
    conf = harness.base_conf(pops)
    conf["UNK_priors"] = "MR"
    got, glog, imp = _run("pop4", conf, lines, "c4")
    assert got["problem"] == ""
    ids = [l.split(",")[0] for l in lines]
    seen = [l.split(",")[0] for l in got["umug"].splitlines()]
    order = {sid: i for i, sid in enumerate(ids)}
    assert [order[s] for s in seen] == sorted(order[s] for s in seen)            # input order kept
    missing = set(ids) - set(seen)
    assert missing == {l.split(",")[1] for l in got["miss"].splitlines()}        # no result <=> .miss
    for name in ("umug", "pmug", "umug_pops", "pmug_pops"):                       # ranks count up from 0, probabilities fall
        last_id, last_rank, last_p = None, -1, None
        for line in got[name].splitlines():
            f = line.rsplit(",", 2)
            sid = line.split(",", 1)[0]
            rank, p = int(f[2]), float(f[1])
            if sid != last_id:
                assert rank == 0
            else:
                assert rank == last_rank + 1 and p <= last_p
            last_id, last_rank, last_p = sid, rank, p
    exp, _ = harness.run_oracle("pop4", conf, lines[:400], tag="c4_orc")
    for k in ("umug", "pmug", "umug_pops", "pmug_pops"):
        n = len(exp[k].splitlines())
        assert got[k].splitlines()[:n] == exp[k].splitlines(), k


def test_config4_full_size_100k():
    """BASELINE config 4 at FULL size (bench.py --workload config4: 100 000 subjects, seed 3): properties over all of it,
    the oracle on a slice from the end, and the first 20 000 subjects equal to what the 20k test's input gives (subjects are
    independent: a bigger batch -- other kernel mix per chunk, table kernels over 27 M pair records -- must not change a row)"""
    rows = synth.read_freqs(synth.CAU_FREQS)
    pops = harness.POPS["pop4"]
    lines = synth.SubjectGen(rows, 3, pops=pops).mixed(100000)
    assert lines[:20000] == synth.SubjectGen(rows, 3, pops=pops).mixed(20000)
    conf = harness.base_conf(pops)
    conf["UNK_priors"] = "MR"
    got, glog, imp = _run("pop4", conf, lines, "c4full")
    assert got["problem"] == "" and not imp.unsupported
    ids = [l.split(",")[0] for l in lines]
    order = {sid: i for i, sid in enumerate(ids)}
    heads = [l.split(",", 1)[0] for l in got["umug"].splitlines() if l.rsplit(",", 1)[1] == "0"]
    assert [order[s] for s in heads] == sorted(order[s] for s in heads) and len(set(heads)) == len(heads)
    assert set(ids) - set(heads) == {l.split(",")[1] for l in got["miss"].splitlines()}
    for name in ("umug", "pmug", "umug_pops", "pmug_pops"):
        last_id, last_rank, last_p = None, -1, None
        for line in got[name].splitlines():
            f = line.rsplit(",", 2)
            sid = line.split(",", 1)[0]
            rank, pr = int(f[2]), float(f[1])
            assert (rank == 0) if sid != last_id else (rank == last_rank + 1 and pr <= last_p)
            last_id, last_rank, last_p = sid, rank, pr
    small, _, _ = _run("pop4", conf, lines[:20000], "c4full20k")
    last20k = ids[19999]
    for k in ("umug", "pmug", "umug_pops", "pmug_pops"):
        big = got[k].splitlines()
        n = len(small[k].splitlines())
        assert big[:n] == small[k].splitlines(), k
        assert n == len(big) or order[big[n].split(",", 1)[0]] > order[last20k], k
    tail = lines[-300:]
    exp, _ = harness.run_oracle("pop4", conf, tail, tag="c4full_orc")
    first_tail = order[tail[0].split(",")[0]]
    for k in ("umug", "pmug", "umug_pops", "pmug_pops"):
        mine = [l for l in got[k].splitlines() if order[l.split(",", 1)[0]] >= first_tail]
        assert mine == exp[k].splitlines(), k


def test_config5_style_high_ambiguity_threshold_1e6():
    """BASELINE config 5 style subjects (8 alternatives per locus and side, number_of_options_threshold
    1e6 -> 32768 candidates per side through the cartesian branch) on the CAU graph."""
    rows = synth.read_freqs(synth.CAU_FREQS)
    lines = synth.SubjectGen(rows, 50).high_ambiguity(3, width=8)
    conf = dict(harness.base_conf(["CAU"]), number_of_options_threshold=1000000)
    _against_oracle("cau", conf, lines, "c5")


@pytest.mark.parametrize("threshold", [200, 30, 8])
def test_label_scan_heavy_workload_vs_oracle(threshold):
    """Low number_of_options_threshold pushes most sides through the label-scan opening: Plan A, Plan B
    and the rewrites of an empty open_phases (impute.py:1619-1627) on that branch."""
    rows = synth.read_freqs(synth.CAU_FREQS)
    lines = synth.SubjectGen(rows, 61).mixed(600, amb=0.7, miss=0.1, recomb=0.5)
    conf = dict(harness.base_conf(["CAU"]), number_of_options_threshold=threshold)
    _against_oracle("cau", conf, lines, "scan%d" % threshold)
    conf4 = dict(harness.base_conf(harness.POPS["pop4"]), number_of_options_threshold=threshold, UNK_priors="MR")
    lines4 = synth.SubjectGen(rows, 62, pops=harness.POPS["pop4"]).mixed(300, amb=0.7, miss=0.1, recomb=0.5)
    _against_oracle("pop4", conf4, lines4, "scan4_%d" % threshold)


def test_empty_and_degenerate_inputs():
    """an empty subject file, a file of blank / unparsable lines only, and a single subject: no device work for the
    first two, same six outputs as the oracle in all cases"""
    from grim.imputation.impute import Imputation
    from grim.imputation.networkx_graph import Graph
    from grim.run_impute_def import load_config

    conf = harness.base_conf(["CAU"])
    for tag, lines in (("blank", [""]), ("only_problem", ["X1,A*01:01", "X2,", "X3"]),
                       ("one", ["D1,A*01:01+A*02:01^B*08:01+B*07:02,CAU,CAU"])):
        _against_oracle("cau", conf, lines, "t_edge_" + tag)
    # zero lines: the reference loops over nothing and leaves six empty files
    work = harness.ensure_graph("cau")
    cwd = os.getcwd()
    os.chdir(work)
    try:
        cfg, _ = load_config("graph_conf.json")
        g = Graph(cfg).build_graph(cfg["node_file"], cfg["top_links_file"], cfg["edges_file"])
        texts = Imputation(g, cfg).impute_lines([], cfg)
    finally:
        os.chdir(cwd)
    assert all(v == "" for v in texts.values()), texts


def test_nine_populations_vs_oracle():
    """P = 9: 81 prior cells, 45 population-pair cells (more than one wave of them), multi-race subjects"""
    rows = synth.read_freqs(synth.CAU_FREQS)
    pops = harness.POPS["pop9"]
    conf = dict(harness.base_conf(pops), UNK_priors="MR")
    lines = synth.SubjectGen(rows, 91, pops=pops).mixed(250, amb=0.3, miss=0.15, recomb=0.2)
    _against_oracle("pop9", conf, lines, "r_pop9")
    conf2 = dict(conf, number_of_pop_results=7, number_of_results=25, UNK_priors="SR")
    _against_oracle("pop9", conf2, synth.SubjectGen(rows, 92, pops=pops).full(150), "r_pop9_full")


def test_crlf_input_file_equals_lf():
    """impute_file hands the file's bytes to the tokenizer; a CRLF file must give what Python's universal
    newlines would have given the reference (the same outputs as the LF file)"""
    from grim.imputation.impute import Imputation
    from grim.imputation.networkx_graph import Graph
    from grim.run_impute_def import load_config

    gname, conf, lines, exp, elog, em = harness.golden("cau_edge")
    work = harness.ensure_graph(gname)
    got_lf, _, _ = _run(gname, conf, lines, "t_crlf_ref", em)
    cwd = os.getcwd()
    os.chdir(work)
    try:
        cfg, out_dir = load_config(os.path.join(work, "conf_t_crlf_ref.json"))
        path = os.path.join(work, "data", "subjects", "t_crlf.csv")
        with open(path, "wb") as fh:
            fh.write(("\r\n".join(lines) + "\r\n").encode())
        cfg["imputation_input_file"] = path
        g = Graph(cfg).build_graph(cfg["node_file"], cfg["top_links_file"], cfg["edges_file"])
        imp = Imputation(g, cfg)
        imp.quiet = True
        imp.impute_file(cfg)
    finally:
        os.chdir(cwd)
    assert harness.read_outputs(work, "t_crlf_ref") == got_lf


def test_timing_mode_and_repeat():
    """grim_batch_set_timing / grim_batch_kernel_ms / grim_batch_run_repeat through the binding: kernel times appear
    only in timing mode, the accumulated mean is the mean, results do not depend on the mode"""
    from grim import _native as nat
    from grim.imputation.impute import Imputation
    from grim.imputation.networkx_graph import Graph
    from grim.run_impute_def import load_config

    work = harness.ensure_graph("cau")
    cwd = os.getcwd()
    os.chdir(work)
    try:
        cfg, _ = load_config("graph_conf.json")
        g = Graph(cfg).build_graph(cfg["node_file"], cfg["top_links_file"], cfg["edges_file"])
        imp = Imputation(g, cfg)
        rows = synth.read_freqs(synth.CAU_FREQS)
        lines = synth.SubjectGen(rows, 5).full(2000) + synth.SubjectGen(rows, 6).mixed(300)
        parsed = nat.Parsed(g.adict, ("\n".join(lines) + "\n").encode(), cfg["planb"])
        priors = np.stack([imp._prior_matrix(r1, r2, cfg["priority"]) for r1, r2 in parsed.races()])
        ctx = nat.default_context(0)
        batch = nat.DeviceBatch(ctx, g.device(ctx), imp._params(cfg, cfg["planb"], False), parsed.subjects(), parsed.tokens(), priors)
        batch.set_timing(False)
        batch.run()
        res0, rows0 = batch.results()
        assert batch.kernel_ms(0) == 0.0
        batch.set_timing(True)
        batch.run_repeat(5)
        res1, rows1 = batch.results()
        assert batch.kernel_ms(3) > 0.0 and batch.kernel_ms(2) > 0.0  # half-wave kernel and Plan B both ran
        total = sum(batch.kernel_ms(w) for w in (3, 5, 9, 4, 2, 6, 7))  # half-wave, one-wave, mid-size, general, Plan B, table kernels, row compaction
        assert batch.kernel_ms(9) > 0.0  # the mid-size kernel took the mixed subjects the one-wave kernel handed on
        assert abs(batch.kernel_ms(0) - total) < 1e-6
        assert batch.kernel_ms(0x10 | 3) > 0.0

        def tables(res, rows):  # row offsets depend on the order in which workgroups took rows; contents must not
            out = []
            for r in res:
                out.append((int(r["status"]), int(r["plan"]), int(r["n_pairs"]), int(r["n_genotypes"]), float(r["max_prob"]),
                            [rows[int(o):int(o) + int(n)].tobytes() for o, n in zip(r["row_off"], r["n_rows"])]))
            return out

        assert tables(res0, rows0) == tables(res1, rows1)
        assert int(res0["n_rows"].sum()) > 0
        batch.close()
    finally:
        os.chdir(cwd)

