"""
Public entry points, same names and signatures as the reference's grim/grim.py:40-87:

    graph_freqs(conf_file="", for_em=False, em_pop=None)
    impute(conf_file="", hap_pop_pair=False, graph=None) -> graph
    impute_instance(config, graph, count_by_prob=None) -> Imputation
    graph_instance(config) -> Graph

`impute` returns (and accepts) the Graph object; it keeps the HBM-resident copy alive so a
second call with `graph=` skips both the CSV load and the upload.
"""

import os
import sys

_PKG_ROOT = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
if _PKG_ROOT not in sys.path:
    sys.path.insert(0, _PKG_ROOT)

from graph_generation import generate_neo4j_multi_hpf  # noqa: E402
from grim.run_impute_def import run_impute  # noqa: E402

from .imputation.impute import Imputation  # noqa: E402
from .imputation.networkx_graph import Graph  # noqa: E402


def _packaged_conf():
    return os.path.join(_PKG_ROOT, "conf", "minimal-configuration.json")


def graph_freqs(conf_file="", for_em=False, em_pop=None):
    use_default_path = False
    if conf_file == "":
        use_default_path = True
        conf_file = _packaged_conf()
    generate_neo4j_multi_hpf.generate_graph(
        config_file=conf_file, em_pop=em_pop, em=for_em, use_default_path=use_default_path,
        quiet=bool(int(os.environ.get("GRIM_QUIET", "0"))))


def impute(conf_file="", hap_pop_pair=False, graph=None):
    project_dir_in_file, project_dir_graph = "", ""
    if conf_file == "":
        conf_file = _packaged_conf()
        project_dir_graph = os.path.join(_PKG_ROOT, "graph_generation") + "/"
        project_dir_in_file = _PKG_ROOT + "/"
    return run_impute(conf_file, project_dir_graph, project_dir_in_file, hap_pop_pair, graph)


def graph_from_freqs(conf_file="", for_em=False, em_pop=None, write_csv=False):
    """graph_freqs() + the graph-loading half of impute() in one step: hpf.csv -> a Graph resident in memory, ready
    for impute(conf, graph=g), without writing and re-reading nodes.csv / edges.csv / top_links.csv (not in the
    reference: there the CSVs are the only interchange).  write_csv=True also leaves the four files as graph_freqs does."""
    from .run_impute_def import load_config

    use_default_path = conf_file == ""
    if use_default_path:
        conf_file = _packaged_conf()
    hpf, pops, cutoffs, loci_map = generate_neo4j_multi_hpf.generator_inputs(conf_file, em_pop, for_em, use_default_path)
    project_dir_graph = os.path.join(_PKG_ROOT, "graph_generation") + "/" if use_default_path else ""
    config, _ = load_config(conf_file, project_dir_graph, _PKG_ROOT + "/" if use_default_path else "")
    paths = None
    if write_csv:
        import json
        import pathlib

        with open(conf_file) as fh:
            raw = json.load(fh)
        csvdir = (os.path.dirname(os.path.realpath(generate_neo4j_multi_hpf.__file__)) + "/" if use_default_path else "") \
            + raw.get("graph_files_path")
        pathlib.Path(csvdir).mkdir(parents=True, exist_ok=True)
        if csvdir[-1] != "/":
            csvdir += "/"
        paths = (csvdir + raw.get("node_csv_file"), csvdir + raw.get("edges_csv_file"), csvdir + raw.get("top_links_csv_file"),
                 csvdir + raw.get("info_node_csv_file"))
    return Graph(config).build_graph_from_hpf(hpf, pops, cutoffs, loci_map, paths)


def impute_instance(config, graph, count_by_prob=None):
    return Imputation(graph, config, count_by_prob)


def graph_instance(config):
    graph = Graph(config)
    graph.build_graph(config["node_file"], config["top_links_file"], config["edges_file"])
    return graph
