"""N>1 path on CPU: world_size-2 and -3 gloo jobs, chunks pulled from the job's store, every rank writing its pieces straight
into the shared output files at the offsets the published chunk sizes imply.
The per-chunk compute is injected (the oracle) -- what is under test is grim/shard.py; the default (HIP) compute under a
process group is tests/test_multi_rank_gpu.py."""
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
if FAIL_RANK == dist.get_rank():
    def compute(cfg, lines, offset):
        raise ValueError("boom on rank %d" % dist.get_rank())
try:
    merged = shard.impute_sharded(CONF, compute=compute, chunk_lines=CHUNK, return_texts=True)
    if dist.get_rank() == 0:
        json.dump(merged, open(OUT, "w"))
except Exception as e:
    open(OUT + ".err%d" % dist.get_rank(), "w").write("%s: %s" % (type(e).__name__, e))
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


def test_chunk_offsets(tmp_path):
    from grim import shard

    for n, tail in ((0, ""), (1, "\n"), (1, ""), (7, "\n"), (16, "\n"), (17, ""), (100, "\n")):
        text = "\n".join("L%d,xx" % i for i in range(n)) + (tail if n else "")
        p = tmp_path / "f.csv"
        p.write_text(text)
        for chunk in (1, 4, 16, 1000):
            offs = shard.chunk_offsets(str(p), chunk)
            parts = [text[offs[c]:offs[c + 1]] for c in range(len(offs) - 1)]
            assert "".join(parts) == text
            assert all(len(x.splitlines()) == chunk for x in parts[:-1]) and (not parts or 0 < len(parts[-1].splitlines()) <= chunk)


def test_chunk_offsets_universal_newlines(tmp_path):
    """lines end as Python's universal-newline open() ends them -- "\n", "\r\n", a lone "\r" -- and at nothing else
    (\x0c and \x85 inside an id are data); chunk starts are line starts, whichever block boundary the "\r\n" straddles"""
    import io

    from grim import shard

    rows = ["A%d,x\x0cy" % i for i in range(23)]
    for sep_cycle in (["\n"], ["\r\n"], ["\r"], ["\n", "\r\n", "\r"]):
        data = "".join(r + sep_cycle[i % len(sep_cycle)] for i, r in enumerate(rows)).encode()
        p = tmp_path / "u.csv"
        p.write_bytes(data)
        want = io.TextIOWrapper(io.BytesIO(data), newline=None).read().split("\n")[:-1]
        for chunk in (1, 3, 7, 50):
            offs = shard.chunk_offsets(str(p), chunk)
            got = []
            for c in range(len(offs) - 1):
                piece = io.TextIOWrapper(io.BytesIO(data[offs[c]:offs[c + 1]]), newline=None).read().split("\n")
                assert piece[-1] == "" and len(piece) - 1 == (chunk if c < len(offs) - 2 else len(rows) - chunk * (len(offs) - 2))
                got += piece[:-1]
            assert got == want


def _launch(tmp_path, world, chunk, lines, port, fail_rank=-1, tag="mr"):
    work = harness.ensure_graph("cau")
    conf = harness.base_conf(["CAU"])
    conf, cpath = harness._write_inputs(work, conf, lines, tag)
    out = str(tmp_path / "merged.json")
    script = tmp_path / "worker.py"
    script.write_text("ROOT=%r\nWORK=%r\nCONF=%r\nOUT=%r\nCHUNK=%d\nFAIL_RANK=%d\n" % (harness.ROOT, work, cpath, out, chunk, fail_rank) + WORKER)
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([harness.PKG, os.path.join(harness.ROOT, "oracle")]),
               MASTER_ADDR="127.0.0.1")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                           "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)], env=env, timeout=600)
    return work, conf, out


def test_world_size_2_gloo(tmp_path):
    """uneven chunks (13 lines each of 95), pulled dynamically by two ranks"""
    rows = synth.read_freqs(synth.CAU_FREQS)
    lines = synth.SubjectGen(rows, 77).mixed(40) + synth.edge_cases("CAU") + synth.SubjectGen(rows, 78).full(41)
    work, conf, out = _launch(tmp_path, 2, 13, lines, 29517)
    merged = json.load(open(out))
    single, _ = harness.run_oracle("cau", conf, lines, tag="mr_single")
    for k in single:
        assert merged[k] == single[k], k
    got = harness.read_outputs(work, "mr")
    for k in single:
        assert got[k] == single[k], "file " + k
    assert not [f for f in os.listdir(os.path.join(work, "output_mr")) if f.startswith(".grim_parts")]


def test_failed_run_then_good_run_in_the_same_directory(tmp_path):
    """a job that dies leaves nothing a later job in the same output directory could pick up: there are no part files (every
    rank writes into the final files, which rank 0 empties first), the store keys carry the job's own id, and part
    directories of older builds lying around are never read"""
    rows = synth.read_freqs(synth.CAU_FREQS)
    lines = synth.SubjectGen(rows, 81).mixed(30) + synth.edge_cases("CAU")[:4]
    work, conf, out = _launch(tmp_path, 2, 5, lines, 29523, fail_rank=1, tag="mrs")
    assert not os.path.exists(out)
    out_dir = os.path.join(work, "output_mrs")
    # what an older build's failed run would have left behind, under the old naming scheme and the new one
    for name in (".grim_parts_1", ".grim_parts_deadbeef0000"):
        os.makedirs(os.path.join(out_dir, name), exist_ok=True)
        for k in ("miss", "umug", "problem"):
            with open(os.path.join(out_dir, name, "%s.%08d" % (k, 3)), "w") as fh:
                fh.write("STALE\n")
            with open(os.path.join(out_dir, name, "%s.rank0" % k), "w") as fh:
                fh.write("STALE\n")
    lines2 = lines[:17]  # a different input: fewer chunks than the stale parts name
    work, conf, out = _launch(tmp_path, 2, 5, lines2, 29525, tag="mrs")
    merged = json.load(open(out))
    single, _ = harness.run_oracle("cau", conf, lines2, tag="mrs_single")
    for k in single:
        assert merged[k] == single[k], k
        assert "STALE" not in merged[k]


def test_world_size_3_more_ranks_than_chunks(tmp_path):
    """two chunks for three ranks: one rank gets no line at all"""
    rows = synth.read_freqs(synth.CAU_FREQS)
    lines = synth.SubjectGen(rows, 79).mixed(30) + synth.edge_cases("CAU")[:6]
    work, conf, out = _launch(tmp_path, 3, 20, lines, 29519, tag="mr3")
    merged = json.load(open(out))
    single, _ = harness.run_oracle("cau", conf, lines, tag="mr3_single")
    for k in single:
        assert merged[k] == single[k], k


def test_a_failing_rank_does_not_hang_the_others(tmp_path):
    """rank 1's compute raises: every rank leaves impute_sharded with an exception instead of waiting forever"""
    rows = synth.read_freqs(synth.CAU_FREQS)
    lines = synth.SubjectGen(rows, 80).full(60)
    work, conf, out = _launch(tmp_path, 2, 5, lines, 29521, fail_rank=1, tag="mrf")
    assert not os.path.exists(out)
    e0, e1 = open(out + ".err0").read(), open(out + ".err1").read()
    assert "rank(s) [1] failed" in e0 and "boom on rank 1" in e0
    assert e1.startswith("ValueError: boom on rank 1")


def test_alone_writes_the_final_files_directly():
    """no process group: one rank pulls every chunk and its part files ARE the outputs (no parts directory, no merge)"""
    import grim_oracle as go
    from grim import shard

    rows = synth.read_freqs(synth.CAU_FREQS)
    lines = synth.SubjectGen(rows, 82).mixed(25) + synth.edge_cases("CAU")[:5]
    work = harness.ensure_graph("cau")
    conf = harness.base_conf(["CAU"])
    conf, cpath = harness._write_inputs(work, conf, lines, "mr1")
    cwd = os.getcwd()
    os.chdir(work)
    try:
        def compute(cfg, chunk, offset):
            ocfg = go.config_from_json(json.load(open(cpath)))
            g = go.OGraph(ocfg["full_loci"]).load(ocfg["node_file"], ocfg["top_links_file"], ocfg["edges_file"])
            texts = go.OracleImputer(g, ocfg).impute_lines(["PAD,,X,X"] * offset + [l.rstrip("\n") for l in chunk])
            texts["problem"] = "".join(l for l in texts["problem"].splitlines(True) if not l.endswith(",PAD\n"))
            return texts

        env_world = os.environ.pop("WORLD_SIZE", None)
        try:
            merged = shard.impute_sharded(cpath, compute=compute, chunk_lines=7, return_texts=True)
        finally:
            if env_world is not None:
                os.environ["WORLD_SIZE"] = env_world
    finally:
        os.chdir(cwd)
    single, _ = harness.run_oracle("cau", conf, lines, tag="mr1_single")
    for k in single:
        assert merged[k] == single[k], k
    assert not [f for f in os.listdir(os.path.join(work, "output_mr1")) if f.startswith(".grim_parts")]
