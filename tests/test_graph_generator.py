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


@pytest.mark.parametrize("name", ["cau", "pop4"])
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
