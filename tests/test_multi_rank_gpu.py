"""The multi-GPU driver with its DEFAULT compute (the HIP engine) under a real process group: two ranks, both on the
one GPU of the test box, chunks pulled from the job's store, outputs merged by rank 0.  impute_sharded joins the job
itself (the worker script never touches torch.distributed)."""
import json
import os
import subprocess
import sys

import pytest

import harness
import synth

pytestmark = pytest.mark.gpu

WORKER = r'''
import json, os, sys
os.environ["GRIM_QUIET"] = "1"
os.chdir(WORK)
from grim import shard
merged = shard.impute_sharded(CONF, chunk_lines=CHUNK, return_texts=True)
if int(os.environ["RANK"]) == 0:
    json.dump(merged, open(OUT, "w"))
'''


def test_two_ranks_default_compute_on_one_gpu(tmp_path):
    work = harness.ensure_graph("pop4")
    rows = synth.read_freqs(synth.CAU_FREQS)
    lines = synth.SubjectGen(rows, 91, pops=harness.POPS["pop4"]).mixed(700) + synth.edge_cases("AFA")
    conf = harness.base_conf(harness.POPS["pop4"])
    conf["UNK_priors"] = "MR"
    conf, cpath = harness._write_inputs(work, conf, lines, "mrg")
    out = str(tmp_path / "merged.json")
    script = tmp_path / "worker.py"
    script.write_text("WORK=%r\nCONF=%r\nOUT=%r\nCHUNK=%d\n" % (work, cpath, out, 100) + WORKER)
    env = dict(os.environ, PYTHONPATH=harness.PKG, MASTER_ADDR="127.0.0.1")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                           "--master-addr", "127.0.0.1", "--master-port", "29531", str(script)], env=env, timeout=900)
    merged = json.load(open(out))
    single, _, _ = harness.run_product("pop4", dict(conf), lines, tag="mrg_single", quiet=True)
    for k in single:
        assert merged[k] == single[k], k
    got = harness.read_outputs(work, "mrg")
    for k in single:
        assert got[k] == single[k], "file " + k
    exp, _ = harness.run_oracle("pop4", conf, lines[:200], tag="mrg_orc")
    n = len(exp["umug"].splitlines())
    assert merged["umug"].splitlines()[:n] == exp["umug"].splitlines()
