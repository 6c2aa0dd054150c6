"""
Multi-GPU driver: one process per GPU, the subject file cut into fixed-size chunks of lines that the ranks PULL from a
shared counter, graph replicated per GPU, NO collective on the data path; every chunk's six outputs go to part files and
rank 0 concatenates them in chunk order.

This is the split the reference's scripts/runfile_mp.py:109-148 intends (`split -l` + one worker per chunk + `cat` of the
per-chunk outputs), with three differences: workers are GPU ranks; chunks are handed out dynamically (a subject's cost
varies by more than 1000x with its ambiguity, so equal line counts are not equal work -- SURVEY 8e); `.miss/.problem`
keep the GLOBAL line index (the reference's per-chunk runs restart at 0).

The only communication is the control plane: an atomic fetch-add on the job's rendezvous store (the next chunk number),
one barrier at the end, and an error slot per rank so that a rank that fails does not leave the others waiting.

Launch:  torchrun --nproc-per-node N --master-addr 127.0.0.1 your_script.py   ->  impute_sharded(conf)
(`impute_sharded` joins the job itself -- gloo, control plane only -- when the caller has not initialised
torch.distributed; alone, without WORLD_SIZE > 1, it runs every chunk in this process).
"""

import os
import pathlib
import shutil
import traceback

OUTPUT_KEYS = ("umug", "umug_pops", "pmug", "pmug_pops", "miss", "problem")
DEFAULT_CHUNK_LINES = 65536


def shard_range(n, rank, world):
    """[begin, end) of rank's contiguous block of ceil(n/world) lines (the static split of runfile_mp.py:113-124)."""
    per = -(-n // world) if world > 0 else n
    return min(n, rank * per), min(n, (rank + 1) * per)


def merge_texts(per_rank):
    """per_rank: list (rank order) of dicts of output texts -> one dict, rank order preserved."""
    return {k: "".join(t.get(k, "") for t in per_rank) for k in OUTPUT_KEYS}


def chunk_offsets(path, chunk_lines):
    """byte offset of every chunk_lines-th line start of the file, plus the file size: chunk c = bytes [off[c], off[c+1]).
    A last line without its newline counts as a line."""
    import numpy as np

    size = os.path.getsize(path)
    if size == 0:
        return [0]
    offs = [0]
    seen = 0  # newlines before the current block
    with open(path, "rb") as fh:
        pos = 0
        while True:
            block = fh.read(1 << 24)
            if not block:
                break
            nl = np.flatnonzero(np.frombuffer(block, dtype=np.uint8) == 10)
            k = chunk_lines - (seen % chunk_lines) - 1  # index (inside nl) of the newline that ends the current chunk
            while k < len(nl):
                offs.append(pos + int(nl[k]) + 1)
                k += chunk_lines
            seen += len(nl)
            pos += len(block)
    if offs[-1] >= size and len(offs) > 1:
        offs.pop()
    offs.append(size)
    return offs


class _Control:
    """the job's control plane: chunk counter, error slots, final barrier"""

    def __init__(self):
        self.rank, self.world, self.dist, self.store = 0, 1, None, None
        self.local = 0
        world_env = int(os.environ.get("WORLD_SIZE", "1"))
        try:
            import torch.distributed as dist
        except ImportError:
            dist = None
        if dist is not None and dist.is_available():
            if not dist.is_initialized() and world_env > 1:
                # the caller did not join the job: do it here (control plane only, so gloo; the GPU work needs no
                # process group at all)
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                dist.init_process_group(backend="gloo")
                self._own_group = True
            if dist.is_initialized():
                self.dist = dist
                self.rank, self.world = dist.get_rank(), dist.get_world_size()
                from torch.distributed import distributed_c10d as c10d

                self.store = c10d._get_default_store()
        if self.dist is None and world_env > 1:
            raise RuntimeError("WORLD_SIZE=%d but torch.distributed is not available: every rank would impute the whole "
                               "file and write the same output files" % world_env)
        self.local = int(os.environ.get("LOCAL_RANK", self.rank))
        self._next = 0

    def next_chunk(self, tag):
        if self.store is None:
            c = self._next
            self._next += 1
            return c
        return int(self.store.add("grim_chunk_" + tag, 1)) - 1

    def report_error(self, tag, text):
        if self.store is not None:
            self.store.set("grim_err_%s_%d" % (tag, self.rank), text)

    def barrier_and_errors(self, tag):
        """-> list of (rank, text) of the ranks that failed (every rank gets the same list)"""
        if self.dist is None:
            return []
        self.store.set("grim_done_%s_%d" % (tag, self.rank), "1")
        self.barrier()
        out = []
        for r in range(self.world):
            key = "grim_err_%s_%d" % (tag, r)
            if self.store.num_keys() and self._has(key):
                out.append((r, self.store.get(key).decode()))
        return out

    def barrier(self):
        if self.dist is None:
            return
        if self.dist.get_backend() == "nccl":  # RCCL wants the rank's device selected before its first collective
            import torch

            n = torch.cuda.device_count()
            if n > 0:
                torch.cuda.set_device(self.local % n)
        self.dist.barrier()

    def _has(self, key):
        try:
            return bool(self.store.check([key]))
        except Exception:  # older stores: no check()
            return False


_run_counter = [0]


def impute_sharded(conf_file, hap_pop_pair=False, graph=None, compute=None, project_dir_graph="",
                   project_dir_in_file="", chunk_lines=None):
    """Run `impute` across the ranks of the torch.distributed job (or alone if there is none).  `compute(config, lines,
    line_offset) -> texts` can be injected (tests); the default runs the HIP engine on this rank's GPU (LOCAL_RANK).
    Returns the merged texts on rank 0 (read back from the files it wrote), None elsewhere; every rank raises when any
    rank failed."""
    from .run_impute_def import load_config

    ctl = _Control()
    _run_counter[0] += 1
    tag = str(_run_counter[0])  # several calls in one job keep separate counters
    chunk_lines = int(chunk_lines or os.environ.get("GRIM_SHARD_LINES", DEFAULT_CHUNK_LINES))
    config, out_dir = load_config(conf_file, project_dir_graph, project_dir_in_file)
    in_path = config["imputation_input_file"]
    parts_dir = os.path.join(out_dir, ".grim_parts_" + tag)
    error = None
    try:
        offs = chunk_offsets(in_path, chunk_lines)
        n_chunks = len(offs) - 1
        if ctl.rank == 0:
            pathlib.Path(out_dir).mkdir(parents=False, exist_ok=True)
        pathlib.Path(parts_dir).mkdir(parents=True, exist_ok=True)
        if compute is None:
            from .imputation.impute import Imputation
            from .imputation.networkx_graph import Graph
            from . import _native as nat

            if graph is None:
                graph = Graph(config).build_graph(config["node_file"], config["top_links_file"], config["edges_file"])
            n_dev = max(1, nat.lib().grim_device_count())
            imp = Imputation(graph, config, device=ctl.local % n_dev)

            def compute(cfg, shard, offset):
                return imp.impute_lines(shard, cfg, em_mr=hap_pop_pair, line_offset=offset, as_bytes=True)

        with open(in_path, "rb") as fh:
            while True:
                c = ctl.next_chunk(tag)
                if c >= n_chunks:
                    break
                fh.seek(offs[c])
                raw = fh.read(offs[c + 1] - offs[c])
                lines = raw.decode().splitlines(True)
                texts = compute(config, lines, c * chunk_lines)
                for k in OUTPUT_KEYS:
                    data = texts.get(k, "")
                    if data:
                        with open(os.path.join(parts_dir, "%s.%08d" % (k, c)), "wb") as out:
                            out.write(data if isinstance(data, bytes) else data.encode())
    except BaseException as e:  # the other ranks must not wait for this one forever: report, reach the barrier, raise
        error = e
        ctl.report_error(tag, "%s: %s\n%s" % (type(e).__name__, e, traceback.format_exc()))
    failed = ctl.barrier_and_errors(tag)
    if error is not None:
        raise error
    if failed:
        raise RuntimeError("impute_sharded: rank(s) %s failed:\n%s" % ([r for r, _ in failed], failed[0][1]))
    merged = None
    if ctl.rank == 0:
        from .imputation.impute import Imputation as _I

        names = {key: (path_key, flag) for key, path_key, flag in _I._OUT_FILES}
        merged = {}
        for k in OUTPUT_KEYS:
            path_key, flag = names[k]
            if flag is not None and not config[flag]:
                merged[k] = ""
                continue
            with open(config[path_key], "wb") as out:  # `cat` of the per-chunk parts in chunk order
                for c in range(n_chunks):
                    part = os.path.join(parts_dir, "%s.%08d" % (k, c))
                    if os.path.exists(part):
                        with open(part, "rb") as src:
                            shutil.copyfileobj(src, out, 1 << 24)
            with open(config[path_key]) as fh:
                merged[k] = fh.read()
        shutil.rmtree(parts_dir, ignore_errors=True)
    ctl.barrier()
    return merged
