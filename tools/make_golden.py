#!/usr/bin/env python3
"""
Generate tests/golden/ by running the REAL reference (py-graph-imputation @ /root/reference)
in this build container.  The reference never travels: only its inputs/outputs (data) are
committed.  Re-run with:  python tools/make_golden.py

Recipe: scratch-copy grim/ + graph_generation/, build the one Cython
module, then for every scenario run produce_hpf -> graph_freqs -> impute exactly as a user
would, from a per-graph work directory.
"""

import contextlib
import hashlib
import io
import json
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import synth  # noqa: E402

REF = "/root/reference"
SCRATCH = "/tmp/grim_ref_scratch"
GOLD = os.path.join(ROOT, "tests", "golden")


def prepare_reference():
    if not os.path.isdir(os.path.join(SCRATCH, "grim")):
        os.makedirs(SCRATCH, exist_ok=True)
        for d in ("grim", "graph_generation"):
            shutil.copytree(os.path.join(REF, d), os.path.join(SCRATCH, d))
        subprocess.check_call(["chmod", "-R", "u+w", SCRATCH])
    so = [f for f in os.listdir(os.path.join(SCRATCH, "grim", "imputation")) if f.startswith("cutils.") and f.endswith(".so")]
    if not so:
        with open(os.path.join(SCRATCH, "setup_cutils.py"), "w") as fh:
            fh.write(
                "from setuptools import setup, Extension\n"
                "from Cython.Build import cythonize\n"
                "setup(ext_modules=cythonize([Extension('grim.imputation.cutils',"
                " ['grim/imputation/cutils.pyx'])], language_level='3'))\n"
            )
        subprocess.check_call([sys.executable, "setup_cutils.py", "build_ext", "--inplace"], cwd=SCRATCH,
                              stdout=subprocess.DEVNULL)
    sys.path.insert(0, SCRATCH)


BASE_CONF = {
    "populations": ["CAU"],
    "freq_trim_threshold": 1e-5,
    "priority": {"alpha": 0.4999999, "eta": 0, "beta": 1e-7, "gamma": 1e-7, "delta": 0.4999999},
    "UNK_priors": "SR",
    "FULL_LOCI": "ABCQR",
    "loci_map": {"A": 1, "B": 2, "C": 3, "DQB1": 4, "DRB1": 5},
    "factor_missing_data": 0.0001,
    "Plan_B_Matrix": [
        [[1, 2, 3, 4, 5]],
        [[1, 2, 3], [4, 5]],
        [[1], [2, 3], [4, 5]],
        [[1, 2, 3], [4], [5]],
        [[1], [2, 3], [4], [5]],
        [[1], [2], [3], [4], [5]],
    ],
    "planb": True,
    "number_of_options_threshold": 100000,
    "epsilon": 1e-3,
    "number_of_results": 10,
    "number_of_pop_results": 100,
    "output_MUUG": True,
    "output_haplotypes": True,
    "freq_data_dir": "data/freqs",
    "freq_file": "output/hpf.csv",
    "graph_files_path": "output/csv/",
    "node_csv_file": "nodes.csv",
    "edges_csv_file": "edges.csv",
    "info_node_csv_file": "info_node.csv",
    "top_links_csv_file": "top_links.csv",
    "imputation_in_file": "data/subjects/input.csv",
    "imputation_out_umug_freq_filename": "don.umug",
    "imputation_out_umug_pops_filename": "don.umug.pops",
    "imputation_out_hap_freq_filename": "don.pmug",
    "imputation_out_hap_pops_filename": "don.pmug.pops",
    "imputation_out_miss_filename": "don.miss",
    "imputation_out_problem_filename": "don.problem",
    "max_haplotypes_number_in_phase": 100,
    "imputation_out_path": "output",
    "pops_count_file": "output/pop_counts_file.txt",
}

POP4 = ["CAU", "AFA", "HIS", "API"]


def md5(path, sort_lines=False):
    with open(path, "rb") as fh:
        data = fh.read()
    if sort_lines:
        lines = data.split(b"\n")
        data = b"\n".join([lines[0]] + sorted(lines[1:]))
    return hashlib.md5(data).hexdigest()


LOCI_MAP_BC = {"A": 1, "B": 3, "C": 2, "DQB1": 4, "DRB1": 5}  # the reference's own default (run_impute_def.py:102-103): B and C swapped


def build_graph(name, pops, loci_map=None):
    """Run the reference's produce_hpf + graph_freqs in SCRATCH/work/<name>."""
    from graph_generation.generate_hpf import produce_hpf
    from grim import grim

    work = os.path.join(SCRATCH, "work", name)
    os.makedirs(os.path.join(work, "data", "freqs"), exist_ok=True)
    os.makedirs(os.path.join(work, "data", "subjects"), exist_ok=True)
    for p in pops:
        shutil.copy(os.path.join(GOLD, "data", "freqs", p + ".freqs.gz"), os.path.join(work, "data", "freqs"))
    conf = dict(BASE_CONF, populations=list(pops))
    if loci_map:
        conf["loci_map"] = dict(loci_map)
    with open(os.path.join(work, "graph_conf.json"), "w") as fh:
        json.dump(conf, fh, indent=1)
    cwd = os.getcwd()
    os.chdir(work)
    try:
        sys.argv = ["x"]
        with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
            produce_hpf("graph_conf.json")
            grim.graph_freqs("graph_conf.json")
    finally:
        os.chdir(cwd)
    gdir = os.path.join(GOLD, "graphs", name)
    os.makedirs(gdir, exist_ok=True)
    csvd = os.path.join(work, "output", "csv")
    info = {
        "populations": list(pops),
        "md5": {
            "hpf.csv": md5(os.path.join(work, "output", "hpf.csv")),
            "pop_counts_file.txt": md5(os.path.join(work, "output", "pop_counts_file.txt")),
            "nodes.csv": md5(os.path.join(csvd, "nodes.csv")),
            "edges.csv": md5(os.path.join(csvd, "edges.csv")),
            "top_links.csv(sorted rows)": md5(os.path.join(csvd, "top_links.csv"), sort_lines=True),
            "info_node.csv": md5(os.path.join(csvd, "info_node.csv")),
        },
        "rows": {f: sum(1 for _ in open(os.path.join(csvd, f))) - 1 for f in ("nodes.csv", "edges.csv", "top_links.csv")},
    }
    with open(os.path.join(gdir, "graph_info.json"), "w") as fh:
        json.dump(info, fh, indent=1)
    shutil.copy(os.path.join(work, "output", "pop_counts_file.txt"), gdir)
    return work


def dump_loader_arrays(name, work, pops):
    """tests/golden/graphs/<name>/loader_arrays.json: what the reference's Graph.build_graph (networkx_graph.py:42-213) holds
    after loading the graph CSVs -- vertex order, the plan-A CSR (Edges / Neighbors_start, sentinel quirk included), the
    plan-B CSR with its connector pseudo-vertices (Whole_*) -- as sizes, SHA-256 digests of the little-endian uint32
    arrays / newline-joined name lists, and the first and last entries."""
    import numpy as np
    from grim.imputation.networkx_graph import Graph as RefGraph
    from grim.run_impute_def import run_impute  # noqa: F401  (imports the module the config helper lives in)

    conf = dict(BASE_CONF, populations=list(pops))
    cwd = os.getcwd()
    os.chdir(work)
    try:
        cfg = {"node_file": "output/csv/nodes.csv", "top_links_file": "output/csv/top_links.csv", "edges_file": "output/csv/edges.csv",
               "full_loci": "".join(sorted(str(v) for v in conf["loci_map"].values())), "nodes_for_plan_A": [], "save_mode": False,
               "pops": list(pops)}
        g = RefGraph(cfg)
        with contextlib.redirect_stdout(io.StringIO()):
            g.build_graph(cfg["node_file"], cfg["top_links_file"], cfg["edges_file"])
    finally:
        os.chdir(cwd)

    def arr(a):
        a = np.ascontiguousarray(np.asarray(a), dtype="<u4")
        return {"n": int(a.size), "sha256": hashlib.sha256(a.tobytes()).hexdigest(), "head": [int(x) for x in a[:8]], "tail": [int(x) for x in a[-8:]]}

    def names(v):
        v = [str(x) for x in v]
        return {"n": len(v), "sha256": hashlib.sha256("\n".join(v).encode()).hexdigest(), "head": v[:3], "tail": v[-3:]}

    V = len(g.Vertices)
    out = {"Vertices": names(g.Vertices), "Edges": arr(g.Edges), "Neighbors_start": arr(g.Neighbors_start),
           "Whole_Vertices_connectors": names(list(g.Whole_Vertices)[V:]), "Whole_Edges": arr(g.Whole_Edges),
           "Whole_Neighbors_start": arr(g.Whole_Neighbors_start), "n_vertices": V, "n_whole_vertices": len(g.Whole_Vertices)}
    with open(os.path.join(GOLD, "graphs", name, "loader_arrays.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print("loader arrays %-6s V=%d Edges=%d Whole_V=%d Whole_Edges=%d" % (name, V, out["Edges"]["n"], out["n_whole_vertices"], out["Whole_Edges"]["n"]))


def run_scenario(name, work, pops, lines, overrides=None, hap_pop_pair=False, bin_masks=None, em=False):
    only = os.environ.get("GOLDEN_ONLY")  # comma-separated scenario names: regenerate just those
    if only and name not in only.split(","):
        return
    from grim import grim

    conf = dict(BASE_CONF, populations=list(pops))
    conf.update(overrides or {})
    for k in [k for k, v in conf.items() if v is None]:  # an override of None removes the key (the reference's default applies)
        del conf[k]
    if bin_masks is not None:
        conf["bin_imputation_in_file"] = "data/subjects/bin.json"
        with open(os.path.join(work, "data", "subjects", "bin.json"), "w") as fh:
            json.dump(bin_masks, fh)
    with open(os.path.join(work, "data", "subjects", "input.csv"), "w") as fh:
        fh.write("\n".join(lines) + "\n")
    with open(os.path.join(work, "conf.json"), "w") as fh:
        json.dump(conf, fh, indent=1)
    for f in ("don.umug", "don.umug.pops", "don.pmug", "don.pmug.pops", "don.miss", "don.problem"):
        p = os.path.join(work, "output", f)
        if os.path.exists(p):
            os.remove(p)
    cwd = os.getcwd()
    os.chdir(work)
    buf = io.StringIO()
    try:
        from grim.imputation.impute import Imputation as _RefImputation

        orig = _RefImputation.impute_file
        if em:  # impute_file(em=True) is only reachable by calling the method; route grim.impute's call to it
            _RefImputation.impute_file = lambda self, config, planb=None, em_mr=False, em=False: orig(self, config, planb, em_mr, True)
        try:
            with contextlib.redirect_stdout(buf), contextlib.redirect_stderr(io.StringIO()):
                grim.impute("conf.json", hap_pop_pair=hap_pop_pair)
        finally:
            _RefImputation.impute_file = orig
    finally:
        os.chdir(cwd)
    out = os.path.join(GOLD, name)
    os.makedirs(out, exist_ok=True)
    shutil.copy(os.path.join(work, "conf.json"), os.path.join(out, "conf.json"))
    shutil.copy(os.path.join(work, "data", "subjects", "input.csv"), os.path.join(out, "input.csv"))
    if bin_masks is not None:
        shutil.copy(os.path.join(work, "data", "subjects", "bin.json"), os.path.join(out, "bin.json"))
    for f in ("don.umug", "don.umug.pops", "don.pmug", "don.pmug.pops", "don.miss", "don.problem"):
        p = os.path.join(work, "output", f)
        if os.path.exists(p):
            shutil.copy(p, os.path.join(out, f))
    log = [l for l in buf.getvalue().splitlines() if "Subject:" in l or l.startswith("in reduce")]
    with open(os.path.join(out, "log.txt"), "w") as fh:
        fh.write("\n".join(log) + "\n")
    meta = {"graph": os.path.basename(work), "hap_pop_pair": hap_pop_pair, "n_subjects": len(lines)}
    if em:
        meta["em"] = True
    with open(os.path.join(out, "meta.json"), "w") as fh:
        json.dump(meta, fh)
    print("scenario %-18s %5d subjects  umug=%d rows" % (
        name, len(lines), sum(1 for _ in open(os.path.join(out, "don.umug"))) if os.path.exists(os.path.join(out, "don.umug")) else -1))


OPEN_GL_CASES = [
    "A*01:01/A*02:01+A*03:01^B*07:02+B*08:01/B*44:02/B*15:01^C*07:01+C*07:02",
    "A*01:01+A*01:01^B*08:01+B*08:01", "A*01:01+A*02:01^^B*08:01+B*07:02", "A*01:01+A*02:01^+^B*08:01+B*07:02",
    "A*01:01+A*02:01", "", " ", "A*01:01", "+A*01:01+A*02:01^B*08:01+B*07:02",
    "A*02:01/A*02:02/A*02:03+A*01:01/A*01:02^B*07:02/B*07:03+B*08:01^C*07:01/C*07:02/C*07:04+C*04:01"
    "^DQB1*02:01+DQB1*03:01/DQB1*03:02^DRB1*03:01+DRB1*04:01",
]


def open_gl_vectors():
    """tests/golden/open_gl_string.json: the reference's Imputation.open_gl_string (the EM hook) on a few GL strings."""
    from grim.imputation.impute import Imputation

    ref = Imputation.__new__(Imputation)  # the method only uses string helpers
    out = []
    for gl in OPEN_GL_CASES:
        for cutoff in (1, 4, 20, 1000):
            try:
                res = ["ok", ref.open_gl_string(gl, cutoff)]
            except Exception as e:
                res = ["exc", type(e).__name__]
            out.append({"gl": gl, "cutoff": cutoff, "result": res})
    with open(os.path.join(GOLD, "open_gl_string.json"), "w") as fh:
        json.dump(out, fh, indent=0)


def main():
    os.environ.setdefault("PYTHONHASHSEED", "0")
    prepare_reference()
    if not os.environ.get("GOLDEN_ONLY") or "open_gl_string" in os.environ["GOLDEN_ONLY"].split(","):
        open_gl_vectors()
    if os.environ.get("GOLDEN_ONLY") == "open_gl_string":
        return
    # ---- data files -----------------------------------------------------------------
    fdir = os.path.join(GOLD, "data", "freqs")
    os.makedirs(fdir, exist_ok=True)
    os.makedirs(os.path.join(GOLD, "data", "subjects"), exist_ok=True)
    shutil.copy(os.path.join(REF, "data", "freqs", "CAU.freqs.gz"), fdir)
    shutil.copy(os.path.join(REF, "data", "subjects", "donor.csv"), os.path.join(GOLD, "data", "subjects"))
    import numpy as np

    cau = synth.read_freqs(os.path.join(fdir, "CAU.freqs.gz"))
    rng = np.random.default_rng(1)
    pop_rows = {"CAU": cau}
    for p in POP4[1:]:
        pop_rows[p] = synth.synth_population(cau, rng)
        synth.write_freqs(os.path.join(fdir, p + ".freqs.gz"), pop_rows[p])

    w1 = build_graph("cau", ["CAU"])
    w4 = build_graph("pop4", POP4)
    if not os.environ.get("GOLDEN_ONLY") or "loader_arrays" in os.environ["GOLDEN_ONLY"].split(","):
        dump_loader_arrays("cau", w1, ["CAU"])
        dump_loader_arrays("pop4", w4, POP4)

    donor = [l.rstrip("\n") for l in open(os.path.join(REF, "data", "subjects", "donor.csv"))]
    run_scenario("cau_min", w1, ["CAU"], donor)
    run_scenario("cau_full", w1, ["CAU"], synth.SubjectGen(cau, 0).full(300))
    run_scenario("cau_mixed", w1, ["CAU"], synth.SubjectGen(cau, 2).mixed(250))
    run_scenario("cau_amb_only", w1, ["CAU"], synth.SubjectGen(cau, 5).mixed(150, amb=0.5, miss=0.0, recomb=0.0))
    run_scenario("cau_edge", w1, ["CAU"], synth.edge_cases("CAU"))
    run_scenario("cau_mr_res1000", w1, ["CAU"], synth.SubjectGen(cau, 7).mixed(60),
                 {"UNK_priors": "MR", "number_of_results": 1000, "number_of_pop_results": 3})
    run_scenario("cau_noplanb", w1, ["CAU"], synth.SubjectGen(cau, 8).mixed(80), {"planb": False})
    run_scenario("cau_muug_only", w1, ["CAU"], synth.SubjectGen(cau, 9).mixed(60), {"output_haplotypes": False})
    run_scenario("cau_haps_only", w1, ["CAU"], synth.SubjectGen(cau, 10).mixed(60), {"output_MUUG": False})
    run_scenario("cau_filter", w1, ["CAU"], synth.SubjectGen(cau, 11).high_ambiguity(12, width=6),
                 {"number_of_options_threshold": 1000})
    run_scenario("cau_top5", w1, ["CAU"], synth.SubjectGen(cau, 12).mixed(60, amb=0.4, miss=0.3),
                 {"max_haplotypes_number_in_phase": 5})
    run_scenario("cau_em_mr", w1, ["CAU"], synth.SubjectGen(cau, 13).mixed(40), hap_pop_pair=True)
    run_scenario("cau_planc", w1, ["CAU"], synth.plan_c_cases("CAU"))
    scan_lines = synth.SubjectGen(cau, 61).mixed(250, amb=0.7, miss=0.1, recomb=0.5)
    run_scenario("cau_scan30", w1, ["CAU"], scan_lines, {"number_of_options_threshold": 30})
    run_scenario("cau_scan8_muug", w1, ["CAU"], scan_lines[:120], {"number_of_options_threshold": 8, "output_haplotypes": False})
    bl = synth.SubjectGen(cau, 15).mixed(120) + synth.SubjectGen(cau, 16).full(80)
    import numpy as _np
    _r = _np.random.default_rng(17)
    masks = {l.split(",")[0]: [int(x) for x in _r.integers(0, 2, 4)] for l in bl[:-3]}  # the last three ids are missing -> .problem
    run_scenario("cau_bin", w1, ["CAU"], bl, bin_masks=masks)

    g4 = synth.SubjectGen(cau, 3, pops=POP4)
    # mix subjects drawn from all four tables
    run_scenario("pop4_mixed", w4, POP4, g4.mixed(300), {"UNK_priors": "MR"})
    run_scenario("pop4_full", w4, POP4, synth.SubjectGen(pop_rows["AFA"], 4, pops=POP4).mixed(150, amb=0.0, miss=0.0, recomb=0.0),
                 {"UNK_priors": "MR"})
    run_scenario("pop4_sr", w4, POP4, synth.SubjectGen(pop_rows["HIS"], 6, pops=POP4).mixed(120), {"UNK_priors": "SR"})
    run_scenario("pop4_edge", w4, POP4, synth.edge_cases("AFA") + synth.edge_cases("UNK"), {"UNK_priors": "MR"})
    run_scenario("pop4_scan30", w4, POP4, synth.SubjectGen(cau, 62, pops=POP4).mixed(200, amb=0.7, miss=0.1, recomb=0.5),
                 {"number_of_options_threshold": 30, "UNK_priors": "MR"})
    run_scenario("pop4_planc", w4, POP4, synth.plan_c_cases("HIS") + synth.plan_c_cases("UNK"), {"UNK_priors": "MR"})
    run_scenario("pop4_planc_haps", w4, POP4, synth.plan_c_cases("API"), {"UNK_priors": "SR", "output_MUUG": False})
    # found by tools/fuzz.py: the MUUG pass ends in Plan C, the phased pass then starts over on the reduced
    # phases and succeeds earlier (impute.py:1637-1654)
    rerun = [l.rstrip("\n") for l in open(os.path.join(HERE, "cases", "planc_rerun.csv"))]
    rerun_conf = {"UNK_priors": "MR", "number_of_options_threshold": 5,
                  "priority": {"alpha": 0.128, "eta": 0.09, "beta": 0.168, "gamma": 0.165, "delta": 0.667}, "epsilon": 1e-7}
    run_scenario("pop4_planc_rerun", w4, POP4, rerun, rerun_conf)
    run_scenario("pop4_planc_rerun_em", w4, POP4, rerun[40:100], rerun_conf, hap_pop_pair=True)
    run_scenario("pop4_planc_wide", w4, POP4, synth.plan_c_wide_cases("AFA") + synth.plan_c_wide_cases("UNK") + rerun[100:140],
                 dict(rerun_conf, UNK_priors="MR"))
    run_scenario("cau_planc_wide_muug", w1, ["CAU"], synth.plan_c_wide_cases("CAU") + synth.plan_c_cases("CAU"),
                 {"number_of_options_threshold": 5, "output_haplotypes": False})
    run_scenario("pop4_planc_em", w4, POP4, synth.plan_c_cases("API") + rerun[:60], dict(rerun_conf, UNK_priors="SR"), em=True)
    run_scenario("pop4_planc_em_haps", w4, POP4, synth.plan_c_cases("HIS"), {"UNK_priors": "MR", "output_MUUG": False}, em=True)
    run_scenario("pop4_em_mr", w4, POP4, synth.SubjectGen(cau, 14, pops=POP4).mixed(40), {"UNK_priors": "MR"},
                 hap_pop_pair=True)
    # epsilon <= 0: call_comp_phase_prob never enters its loop and returns its {"Haps": "NaN"} sentinel (impute.py:1663-1665);
    # the writers trip over the missing "Pops" key -> raw line in .problem (and .miss when only haplotypes are written)
    eps_lines = synth.edge_cases("CAU") + synth.SubjectGen(cau, 18).mixed(25) + synth.plan_c_cases("CAU")[:3]
    run_scenario("cau_eps0", w1, ["CAU"], eps_lines, {"epsilon": 0})
    run_scenario("cau_eps0_haps", w1, ["CAU"], eps_lines, {"epsilon": 0.0, "output_MUUG": False})
    run_scenario("cau_epsneg_muug", w1, ["CAU"], eps_lines, {"epsilon": -1e-3, "output_haplotypes": False})
    run_scenario("cau_eps0_noplanb", w1, ["CAU"], eps_lines, {"epsilon": 0, "planb": False})
    # a loci_map whose index order is not the alphabetical locus order -- the reference's own default map, B = 3 and C = 2 --
    # used for graph generation and imputation alike: graph names are A~C~B~DQB1~DRB1, a subject's candidate names are its
    # alleles in sorted order (impute.py:271), so cartesian candidates that hold both B and C never match; label-scan
    # candidates (graph order) do
    wbc = build_graph("cau_bc", ["CAU"], LOCI_MAP_BC)
    w4bc = build_graph("pop4_bc", POP4, LOCI_MAP_BC)
    bc = {"loci_map": LOCI_MAP_BC}
    run_scenario("bc_cau_mixed", wbc, ["CAU"], synth.SubjectGen(cau, 31).mixed(200) + synth.edge_cases("CAU"), bc)
    run_scenario("bc_cau_scan30", wbc, ["CAU"], synth.SubjectGen(cau, 32).mixed(150, amb=0.7, miss=0.1, recomb=0.5),
                 dict(bc, number_of_options_threshold=30))
    run_scenario("bc_pop4_mixed", w4bc, POP4, synth.SubjectGen(cau, 33, pops=POP4).mixed(200) + synth.plan_c_cases("HIS"),
                 dict(bc, UNK_priors="MR"))
    # the same graph, and a conf WITHOUT a loci_map: run_impute_def.py:102-103 supplies this very map
    run_scenario("bc_cau_default_map", wbc, ["CAU"], synth.SubjectGen(cau, 35).mixed(120, amb=0.4, miss=0.3, recomb=0.4) + synth.plan_c_cases("CAU"),
                 {"loci_map": None})
    run_scenario("bc_pop4_scan8", w4bc, POP4, synth.SubjectGen(cau, 34, pops=POP4).mixed(120, amb=0.7, miss=0.2, recomb=0.5),
                 dict(bc, UNK_priors="MR", number_of_options_threshold=8))


    # save_space_mode (open_option_ cuts either operand to its ten largest entries, impute.py:1048-1059).  The subject lines of
    # these two are the committed tests/golden/<name>/input.csv (mixed + plan_c_cases + edge cases / plan_c_wide_cases; the
    # generator call of the round they were made in was not kept): the outputs are re-made from them.
    def committed_input(name):
        return [l.rstrip("\n") for l in open(os.path.join(GOLD, name, "input.csv"))]

    run_scenario("cau_save", w1, ["CAU"], committed_input("cau_save"), {"save_space_mode": True})
    run_scenario("pop4_save", w4, POP4, committed_input("pop4_save"), {"UNK_priors": "MR", "save_space_mode": True})
    # GL strings that name a locus twice or mix loci in an entry: what the reference makes of them (this build reports such
    # subjects -- reason 8 -- instead of answering)
    irr = synth.irregular_cases("CAU")
    mix = synth.SubjectGen(cau, 41).mixed(30)
    irr_lines = [x for pair in zip(mix[:10], irr) for x in pair] + mix[10:]
    run_scenario("cau_irregular", w1, ["CAU"], irr_lines)
    run_scenario("cau_irregular_noplanb", w1, ["CAU"], irr_lines, {"planb": False})
    irr4 = synth.irregular_cases("HIS")
    mix4 = synth.SubjectGen(cau, 42, pops=POP4).mixed(30)
    run_scenario("pop4_irregular", w4, POP4, [x for pair in zip(mix4[:10], irr4) for x in pair] + mix4[10:], {"UNK_priors": "MR"})


if __name__ == "__main__":
    main()
