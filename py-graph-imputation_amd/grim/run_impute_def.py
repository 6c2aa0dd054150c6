"""
run_impute(conf_file, project_dir_graph, project_dir_in_file, hap_pop_pair, graph): conf JSON ->
config dict -> graph (built or reused) -> Imputation.impute_file.

Drop-in for the reference's grim/run_impute_def.py:41-211: same conf keys and defaults
(:63-129), same banner (:132-186), same `full_loci` derivation (:188-192), same output
directory rule (`mkdir(parents=False)`, :202).
"""

import json
import os
import pathlib
from pathlib import Path

from .imputation.impute import Imputation
from .imputation.networkx_graph import Graph

DEFAULT_PLAN_B_MATRIX = [
    [[1, 2, 3, 4, 5]],
    [[1, 2, 3], [4, 5]],
    [[1], [2, 3], [4, 5]],
    [[1, 2, 3], [4], [5]],
    [[1], [2, 3], [4], [5]],
    [[1], [2], [3], [4], [5]],
]


def full_path(output, original_path):
    """<dir of original_path>/<output>/<file name of original_path>  (run_impute_def.py:19-38)"""
    p = Path(original_path)
    return str(p.parent / output / p.name)


def load_config(conf_file, project_dir_graph="", project_dir_in_file=""):
    with open(conf_file) as fh:
        js = json.load(fh)
    gpath = js.get("graph_files_path")
    if gpath[-1] != "/":
        gpath += "/"
    out_dir = js.get("imputation_out_path", "output")
    if out_dir[-1] != "/":
        out_dir += "/"
    g = project_dir_graph
    config = {
        "planb": js.get("planb", True),
        "pops": js.get("populations"),
        "priority": js.get("priority"),
        "epsilon": js.get("epsilon", 1e-3),
        "number_of_results": js.get("number_of_results", 1000),
        "number_of_pop_results": js.get("number_of_pop_results", 100),
        "output_MUUG": js.get("output_MUUG", True),
        "output_haplotypes": js.get("output_haplotypes", False),
        "node_file": g + gpath + js.get("node_csv_file"),
        "top_links_file": g + gpath + js.get("top_links_csv_file"),
        "edges_file": g + gpath + js.get("edges_csv_file"),
        "imputation_input_file": project_dir_in_file + js.get("imputation_in_file"),
        "factor_missing_data": js.get("factor_missing_data", 0.01),
        "loci_map": js.get("loci_map", {"A": 1, "B": 3, "C": 2, "DQB1": 4, "DRB1": 5}),
        "matrix_planb": js.get("Plan_B_Matrix", DEFAULT_PLAN_B_MATRIX),
        "pops_count_file": g + js.get("pops_count_file", ""),
        "use_pops_count_file": js.get("pops_count_file", False),
        "number_of_options_threshold": js.get("number_of_options_threshold", 100000),
        "max_haplotypes_number_in_phase": js.get("max_haplotypes_number_in_phase", 100),
        "bin_imputation_input_file": project_dir_in_file + js.get("bin_imputation_in_file", "None"),
        "nodes_for_plan_A": js.get("Plan_A_Matrix", []),
        "save_mode": js.get("save_space_mode", False),
        "UNK_priors": js.get("UNK_priors", "MR"),
    }
    for key, name in (
        ("imputation_out_umug_freq_file", "imputation_out_umug_freq_filename"),
        ("imputation_out_umug_pops_file", "imputation_out_umug_pops_filename"),
        ("imputation_out_hap_freq_file", "imputation_out_hap_freq_filename"),
        ("imputation_out_hap_pops_file", "imputation_out_hap_pops_filename"),
        ("imputation_out_miss_file", "imputation_out_miss_filename"),
        ("imputation_out_problem_file", "imputation_out_problem_filename"),
    ):
        config[key] = full_path(out_dir, js.get(name))
    config["full_loci"] = "".join(sorted({str(v) for v in config["loci_map"].values()}))
    return config, out_dir


def print_banner(config):
    bar = "*" * 100
    rows = [
        ("Population", "pops"), ("Priority", "priority"), ("UNK priority", "UNK_priors"), ("Epsilon", "epsilon"),
        ("Plan B", "planb"), ("Number of Results", "number_of_results"),
        ("Number of Population Results", "number_of_pop_results"), ("Nodes File", "node_file"),
        ("Top Links File", "edges_file"), ("Input File", "imputation_input_file"),
        ("Output UMUG Format", "output_MUUG"), ("Output UMUG Freq Filename", "imputation_out_umug_freq_file"),
        ("Output UMUG Pops Filename", "imputation_out_umug_pops_file"), ("Output Haplotype Format", "output_haplotypes"),
        ("Output HAP Freq Filename", "imputation_out_hap_freq_file"),
        ("Output HAP Pops Filename", "imputation_out_hap_pops_file"), ("Output Miss Filename", "imputation_out_miss_file"),
        ("Output Problem Filename", "imputation_out_problem_file"), ("Factor Missing Data", "factor_missing_data"),
        ("Loci Map", "loci_map"), ("Plan B Matrix", "matrix_planb"), ("Pops Count File", "pops_count_file"),
        ("Use Pops Count File", "use_pops_count_file"), ("Number of Options Threshold", "number_of_options_threshold"),
        ("Max Number of haplotypes in phase", "max_haplotypes_number_in_phase"),
    ]
    print(bar)
    print("Performing imputation based on:")
    for label, key in rows:
        print("\t{}: {}".format(label, config[key]))
    if config["nodes_for_plan_A"]:
        print("\tNodes in plan A: {}".format(config["nodes_for_plan_A"]))
    print("\tSave space mode: {}".format(config["save_mode"]))
    print(bar)


def run_impute(conf_file="../conf/minimal-configuration.json", project_dir_graph="", project_dir_in_file="",
               hap_pop_pair=False, graph=None):
    config, out_dir = load_config(conf_file, project_dir_graph, project_dir_in_file)
    if not int(os.environ.get("GRIM_QUIET", "0")):
        print_banner(config)
    if graph is None:
        graph = Graph(config)
        graph.build_graph(config["node_file"], config["top_links_file"], config["edges_file"])
    imputation = Imputation(graph, config)
    pathlib.Path(out_dir).mkdir(parents=False, exist_ok=True)
    imputation.impute_file(config, em_mr=hap_pop_pair)
    return graph
