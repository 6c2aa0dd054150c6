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


WORKER_UNSUP = r'''
import json, os, sys
os.environ["GRIM_QUIET"] = "1"
os.environ["GRIM_ON_UNSUPPORTED"] = MODE
os.chdir(WORK)
from grim import shard
rank = int(os.environ["RANK"])
try:
    merged = shard.impute_sharded(CONF, chunk_lines=CHUNK, return_texts=True)
    if rank == 0:
        json.dump(merged, open(OUT, "w"))
except Exception as e:
    open(OUT + ".err%d" % rank, "w").write("%s: %s" % (type(e).__name__, e))
'''


def _launch_unsup(tmp_path, mode, port):
    work = harness.ensure_graph("cau")
    rows = synth.read_freqs(synth.CAU_FREQS)
    good = synth.SubjectGen(rows, 92).mixed(300)
    irr = synth.irregular_cases("CAU")
    lines = good[:120] + irr[:2] + good[120:260] + irr[2:4] + good[260:]  # the reported subjects fall into different chunks
    conf = harness.base_conf(["CAU"])
    conf, cpath = harness._write_inputs(work, conf, lines, "mru_" + mode)
    out = str(tmp_path / ("merged_%s.json" % mode))
    script = tmp_path / ("worker_%s.py" % mode)
    script.write_text("WORK=%r\nCONF=%r\nOUT=%r\nCHUNK=%d\nMODE=%r\n" % (work, cpath, out, 50, mode) + WORKER_UNSUP)
    env = dict(os.environ, PYTHONPATH=harness.PKG, MASTER_ADDR="127.0.0.1")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                           "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)], env=env, timeout=900)
    return conf, lines, out


def test_sharded_job_raises_on_unsupported_subjects_like_impute_file(tmp_path):
    """a subject the device cannot take (here: a GL string that names a locus twice, reason 8) fails the sharded job on EVERY
    rank in the default mode, exactly as the single-GPU impute_file raises -- it is never silently left out"""
    conf, lines, out = _launch_unsup(tmp_path, "raise", 29541)
    assert not os.path.exists(out)
    errs = [open(out + ".err%d" % r).read() for r in range(2)]
    assert any(e.startswith("UnsupportedSubjects") for e in errs)
    assert all(e.startswith("UnsupportedSubjects") or "failed" in e for e in errs)


def test_sharded_job_skip_mode_returns_the_subjects_of_all_ranks(tmp_path):
    """GRIM_ON_UNSUPPORTED=skip: the job goes on without them and rank 0 returns every rank's list, by global line"""
    conf, lines, out = _launch_unsup(tmp_path, "skip", 29543)
    merged = json.load(open(out))
    want = [(i, l.split(",")[0], 8) for i, l in enumerate(lines) if l.split(",")[0] in ("I0", "I1", "I2", "I3")]
    assert [tuple(u) for u in merged["unsupported"]] == want
    single, _, imp = harness.run_product("cau", dict(conf), lines, tag="mru_single", quiet=True, on_unsupported="skip")
    assert [tuple(u) for u in imp.unsupported] == want
    for k in single:
        assert merged[k] == single[k], k


def test_bench_gpus_flag_starts_its_ranks(tmp_path):
    """`python bench.py --gpus 2` with no torchrun around it: the script starts its two ranks itself (children; both on the
    box's one GPU here) and rank 0's JSON line says n_gpus 2 and names both ranks"""
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(harness.ROOT, "bench.py"), "--gpus", "2", "--workload", "config4", "--subjects", "2000",
                        "--steps", "2", "--warmup", "1", "--min-seconds", "0.1", "--kernel-steps", "2", "--no-file", "--no-cpu-baseline"],
                       env=env, stdout=subprocess.PIPE, timeout=900)
    assert p.returncode == 0
    line = [l for l in p.stdout.decode().splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["value"] > 0
    assert sorted(r["rank"] for r in out["config"]["ranks"]) == [0, 1]
    # and a job whose size contradicts the flag is refused
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p2 = subprocess.run([sys.executable, os.path.join(harness.ROOT, "bench.py"), "--gpus", "2"], env=env2, stdout=subprocess.PIPE,
                        stderr=subprocess.PIPE, timeout=300)
    assert p2.returncode != 0 and b"WORLD_SIZE=1" in p2.stderr
