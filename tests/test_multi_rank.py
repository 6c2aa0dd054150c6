"""N>1 path on CPU: world_size-2 gloo job, contiguous shards, ordered gather on rank 0.
The per-shard compute is injected (the oracle) -- what is under test is grim/shard.py."""
import json
import os
import subprocess
import sys

import harness
import synth

WORKER = r'''
import json, os, sys
sys.path.insert(0, os.path.join(ROOT, "tools")); import harness
import torch.distributed as dist
import grim_oracle as go
from grim import shard
dist.init_process_group(backend="gloo")
os.chdir(WORK)
def compute(cfg, lines, offset):
    ocfg = go.config_from_json(json.load(open(CONF)))
    g = go.OGraph(ocfg["full_loci"]).load(ocfg["node_file"], ocfg["top_links_file"], ocfg["edges_file"])
    imp = go.OracleImputer(g, ocfg)
    pad = ["PAD,,X,X"] * offset            # keep the oracle's line numbering global
    texts = imp.impute_lines(pad + [l.rstrip("\n") for l in lines])
    texts["problem"] = "".join(l for l in texts["problem"].splitlines(True) if not l.endswith(",PAD\n"))
    return texts
merged = shard.impute_sharded(CONF, compute=compute)
if dist.get_rank() == 0:
    json.dump(merged, open(OUT, "w"))
dist.barrier(); dist.destroy_process_group()
'''


def test_shard_ranges():
    from grim import shard

    for n in (0, 1, 7, 8, 9, 10000):
        for world in (1, 2, 3, 8):
            cover = []
            for r in range(world):
                lo, hi = shard.shard_range(n, r, world)
                cover.extend(range(lo, hi))
            assert cover == list(range(n))


def test_world_size_2_gloo(tmp_path):
    work = harness.ensure_graph("cau")
    rows = synth.read_freqs(synth.CAU_FREQS)
    lines = synth.SubjectGen(rows, 77).mixed(40) + synth.edge_cases("CAU") + synth.SubjectGen(rows, 78).full(41)
    conf = harness.base_conf(["CAU"])
    conf, cpath = harness._write_inputs(work, conf, lines, "mr")
    out = str(tmp_path / "merged.json")
    script = tmp_path / "worker.py"
    script.write_text("ROOT=%r\nWORK=%r\nCONF=%r\nOUT=%r\n" % (harness.ROOT, work, cpath, out) + WORKER)
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([harness.PKG, os.path.join(harness.ROOT, "oracle")]),
               MASTER_ADDR="127.0.0.1")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                           "--master-addr", "127.0.0.1", "--master-port", "29517", str(script)], env=env, timeout=600)
    merged = json.load(open(out))
    single, _ = harness.run_oracle("cau", conf, lines, tag="mr_single")
    for k in single:
        assert merged[k] == single[k], k
    got = harness.read_outputs(work, "mr")
    for k in single:
        assert got[k] == single[k], "file " + k
