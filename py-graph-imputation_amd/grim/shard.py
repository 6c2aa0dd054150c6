"""
Multi-GPU driver: one process per GPU, the subject file cut into fixed-size chunks of lines that the ranks PULL from a
shared counter, graph replicated per GPU, NO collective on the data path.

This is the split the reference's scripts/runfile_mp.py:109-148 intends (`split -l` + one worker per chunk + `cat` of the
per-chunk outputs), with three differences: workers are GPU ranks; chunks are handed out dynamically (a subject's cost
varies by more than 1000x with its ambiguity, so equal line counts are not equal work -- SURVEY 8e); `.miss/.problem`
keep the GLOBAL line index (the reference's per-chunk runs restart at 0).

Data path of a rank: ONE long-lived streaming pipeline (grim_stream: tokenizer threads -> device -> formatter threads ->
ordered pwrite) for the whole job.  A pulled chunk is a byte range of the input file; its raw bytes go to the stream as one
input SEGMENT (grim_stream_segment carries the chunk's global line index), so chunk k+1 is tokenised while chunk k is on
the device -- no per-chunk stream, no Python line splitting.  The rank's six outputs are six PART FILES that grow in the
order the rank pulled its chunks; a manifest records, per chunk, where its piece of every part file ends.  Rank 0 then
assembles the final files in chunk order with copy_file_range (the kernel moves the bytes; nothing is read back into
Python).

The only communication is the control plane: an atomic fetch-add on the job's rendezvous store (the next chunk number),
the job's unique parts directory name, one barrier at the end, and an error slot per rank so that a rank that fails does
not leave the others waiting.

Launch:  torchrun --nproc-per-node N --master-addr 127.0.0.1 your_script.py   ->  impute_sharded(conf)
(`impute_sharded` joins the job itself -- gloo, control plane only -- when the caller has not initialised
torch.distributed; alone, without WORLD_SIZE > 1, it runs every chunk in this process).
"""

import json
import os
import pathlib
import shutil
import traceback
import uuid

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
    Lines end the way Python's universal-newline open() -- and the library's grim_stream_write_text -- ends them: at "\\n",
    at "\\r\\n" (one end, after the "\\n") and at a lone "\\r".  A last line without its line end counts as a line.
    (The library scans the file: grim_chunk_offsets, csrc/grim_stream.cpp.)"""
    from . import _native as nat

    return nat.chunk_offsets(path, chunk_lines)


class _Control:
    """the job's control plane: chunk counter, shared values, error slots, final barrier"""

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

    def share(self, key, make):
        """rank 0's make() for every rank (the store's get() blocks until rank 0 has set the key)"""
        if self.store is None:
            return make()
        if self.rank == 0:
            v = make()
            self.store.set(key, v)
            return v
        return self.store.get(key).decode()

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


class _StreamSink:
    """a rank's data path: one grim_stream for the whole job, a chunk = one input segment, outputs = six part files"""

    def __init__(self, imp, config, hap_pop_pair, part_paths, flags):
        from . import _native as nat

        self.nat = nat
        planb = config["planb"]
        params = imp._params(config, planb, hap_pop_pair, False)
        ps, keep = nat.prior_spec(config["priority"], imp.unk_priors, imp.count_by_prob)
        ctx = nat.default_context(imp.device)
        out_paths = {k: part_paths[k] for k in OUTPUT_KEYS if flags[k]}
        self._keep = (params, ps, keep)
        self.imp = imp
        self.st = nat.Stream(ctx, imp.netGraph.device(ctx), imp.netGraph.adict, params, ps, imp.populations,
                             out_paths=out_paths, want_log=False, masks=imp._phase_masks(config))
        self.first = True

    def feed(self, raw, line_offset):
        if not self.first or line_offset:
            self.st.segment(line_offset)  # (the first chunk is segment 0 unless it does not start at line 0)
        self.first = False
        self.st.write_text(raw)

    def finish(self):
        """-> per fed chunk, cumulative bytes of the six texts at its end"""
        try:
            self.st.finish()
            ends = self.st.segment_ends()
            self.imp.unsupported = self.st.unsupported()
        finally:
            self.st.close()
        return [e[:6] for e in ends]

    def abort(self):
        try:
            self.st.close()
        except Exception:
            pass


class _ComputeSink:
    """the same layout from an injected per-chunk compute(config, lines, line_offset) -> texts (tests: the oracle)"""

    def __init__(self, compute, config, part_paths, flags):
        self.compute, self.config = compute, config
        self.fh = {k: open(part_paths[k], "wb") for k in OUTPUT_KEYS if flags[k]}
        self.pos = {k: 0 for k in OUTPUT_KEYS}
        self.ends = [[0] * 6]  # segment 0 (empty until the first chunk without an offset arrives)
        self.first = True

    def feed(self, raw, line_offset):
        import io

        # universal newlines, as the product path's grim_stream_write_text
        text = io.TextIOWrapper(io.BytesIO(raw), encoding="utf-8", newline=None).read()
        lines = [l + "\n" for l in text.split("\n")]
        if text.endswith("\n") or not text:
            lines.pop()  # the text ended with a line end: what follows it is not a line
        texts = self.compute(self.config, lines, line_offset)
        for k in OUTPUT_KEYS:
            data = texts.get(k, "")
            if data and k in self.fh:
                data = data if isinstance(data, bytes) else data.encode()
                self.fh[k].write(data)
                self.pos[k] += len(data)
        if self.first and not line_offset:
            self.ends[0] = [self.pos[k] for k in OUTPUT_KEYS]
        else:
            self.ends.append([self.pos[k] for k in OUTPUT_KEYS])
        self.first = False

    def finish(self):
        for fh in self.fh.values():
            fh.close()
        return self.ends

    def abort(self):
        for fh in self.fh.values():
            try:
                fh.close()
            except Exception:
                pass


def _copy_range(src_fd, dst_fd, off, n, dst_off):
    """n bytes of src from offset off to dst at offset dst_off: copy_file_range where the kernel offers it (the bytes never
    enter this process), pread / pwrite otherwise"""
    while n > 0:
        try:
            k = os.copy_file_range(src_fd, dst_fd, n, offset_src=off, offset_dst=dst_off)
        except (AttributeError, OSError):
            k = -1
        if k <= 0:
            buf = os.pread(src_fd, min(n, 1 << 24), off)
            if not buf:
                raise IOError("part file shorter than its manifest says")
            os.pwrite(dst_fd, buf, dst_off)
            k = len(buf)
        off += k
        dst_off += k
        n -= k


def impute_sharded(conf_file, hap_pop_pair=False, graph=None, compute=None, project_dir_graph="",
                   project_dir_in_file="", chunk_lines=None, return_texts=False):
    """Run `impute` across the ranks of the torch.distributed job (or alone if there is none).  `compute(config, lines,
    line_offset) -> texts` can be injected (tests); the default runs the HIP engine on this rank's GPU (LOCAL_RANK) as one
    stream per rank.  Rank 0 returns {key: path of the final file} (the texts themselves with return_texts=True: a test
    convenience -- the product path never reads its outputs back), the other ranks None; every rank raises when any rank
    failed."""
    from .run_impute_def import load_config
    from .imputation.impute import Imputation as _I

    ctl = _Control()
    chunk_lines = int(chunk_lines or os.environ.get("GRIM_SHARD_LINES", DEFAULT_CHUNK_LINES))
    names = {key: (path_key, flag) for key, path_key, flag in _I._OUT_FILES}
    error = None
    sink = None
    parts_dir = None
    alone = ctl.world == 1  # nothing to merge: the one rank's part files ARE the outputs
    # the job's id: unique per call and per job, published by rank 0 -- a parts directory left behind by a run that died
    # can never be taken for this run's (and the manifests, not a directory listing, say what gets merged)
    tag = ctl.share("grim_job_%d" % _next_call(), lambda: uuid.uuid4().hex[:12])
    try:
        config, out_dir = load_config(conf_file, project_dir_graph, project_dir_in_file)
        in_path = config["imputation_input_file"]
        flags = {k: (names[k][1] is None or bool(config[names[k][1]])) for k in OUTPUT_KEYS}
        if ctl.rank == 0:
            pathlib.Path(out_dir).mkdir(parents=False, exist_ok=True)
        if alone:
            part_paths = {k: config[names[k][0]] for k in OUTPUT_KEYS}
        else:
            parts_dir = os.path.join(out_dir, ".grim_parts_" + tag)
            pathlib.Path(parts_dir).mkdir(parents=True, exist_ok=True)
            part_paths = {k: os.path.join(parts_dir, "%s.rank%d" % (k, ctl.rank)) for k in OUTPUT_KEYS}
        offs = chunk_offsets(in_path, chunk_lines)
        n_chunks = len(offs) - 1
        if compute is None:
            from .imputation.networkx_graph import Graph
            from . import _native as nat

            if graph is None:
                graph = Graph(config).build_graph(config["node_file"], config["top_links_file"], config["edges_file"])
            n_dev = max(1, nat.lib().grim_device_count())
            imp = _I(graph, config, device=ctl.local % n_dev)
            sink = _StreamSink(imp, config, hap_pop_pair, part_paths, flags)
        else:
            sink = _ComputeSink(compute, config, part_paths, flags)
        mine = []  # chunks this rank pulled, in order
        with open(in_path, "rb") as fh:
            while True:
                c = ctl.next_chunk(tag)
                if c >= n_chunks:
                    break
                fh.seek(offs[c])
                sink.feed(fh.read(offs[c + 1] - offs[c]), c * chunk_lines)
                mine.append(c)
        ends = sink.finish()
        sink = None
        # segment 0 exists even when the first chunk opened a new segment (it did unless it was chunk 0): drop the empty one
        if len(ends) == len(mine) + 1:
            ends = ends[1:]
        if len(ends) != len(mine):
            raise RuntimeError("internal: %d segments for %d chunks" % (len(ends), len(mine)))
        if not alone:
            with open(os.path.join(parts_dir, "manifest.rank%d.json" % ctl.rank), "w") as fh:
                json.dump({"chunks": mine, "ends": ends}, fh)
    except BaseException as e:  # the other ranks must not wait for this one forever: report, reach the barrier, raise
        error = e
        if sink is not None:
            sink.abort()
        ctl.report_error(tag, "%s: %s\n%s" % (type(e).__name__, e, traceback.format_exc()))
    failed = ctl.barrier_and_errors(tag)
    result = None
    try:
        if error is not None:
            raise error
        if failed:
            raise RuntimeError("impute_sharded: rank(s) %s failed:\n%s" % ([r for r, _ in failed], failed[0][1]))
        if not alone:
            # every rank reads every manifest: where chunk c's piece of every output sits in its rank's part file, hence
            # where it belongs in the final file (the sizes of the chunks before it) -- and moves ITS OWN pieces there, so the
            # merge is as parallel as the job.  Rank 0 creates the files at their final size first.
            where = {}  # chunk -> (rank, start offsets in the part file, end offsets)
            for r in range(ctl.world):
                with open(os.path.join(parts_dir, "manifest.rank%d.json" % r)) as fh:
                    m = json.load(fh)
                prev = [0] * 6
                for c, e in zip(m["chunks"], m["ends"]):
                    where[c] = (r, prev, e)
                    prev = e
            missing = [c for c in range(n_chunks) if c not in where]
            if missing:
                raise RuntimeError("impute_sharded: no rank reported chunk(s) %s" % missing[:8])
            final_off = [[0] * 6]  # final_off[c][k]: where chunk c's piece of output k starts in the final file
            for c in range(n_chunks):
                r, a, e = where[c]
                final_off.append([final_off[-1][k] + e[k] - a[k] for k in range(6)])
            if ctl.rank == 0:
                for ki, k in enumerate(OUTPUT_KEYS):
                    if flags[k]:
                        fd = os.open(config[names[k][0]], os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
                        try:
                            os.ftruncate(fd, final_off[n_chunks][ki])
                        finally:
                            os.close(fd)
            ctl.barrier()
            for ki, k in enumerate(OUTPUT_KEYS):
                if not flags[k]:
                    continue
                pieces = [(c, where[c]) for c in range(n_chunks) if where[c][0] == ctl.rank and where[c][2][ki] > where[c][1][ki]]
                if not pieces:
                    continue
                src = os.open(os.path.join(parts_dir, "%s.rank%d" % (k, ctl.rank)), os.O_RDONLY)
                dst = os.open(config[names[k][0]], os.O_WRONLY)
                try:
                    for c, (r, a, e) in pieces:
                        _copy_range(src, dst, a[ki], e[ki] - a[ki], final_off[c][ki])
                finally:
                    os.close(src)
                    os.close(dst)
        if ctl.rank == 0:
            result = {}
            for k in OUTPUT_KEYS:
                path_key, _ = names[k]
                if not flags[k]:
                    result[k] = "" if return_texts else None
                elif not return_texts:
                    result[k] = config[path_key]
    finally:
        ctl.barrier()  # nobody removes the parts before every rank is through with them
        if ctl.rank == 0 and parts_dir is not None:
            shutil.rmtree(parts_dir, ignore_errors=True)
    if ctl.rank == 0 and return_texts and result is not None:
        for k in OUTPUT_KEYS:
            if flags[k]:
                with open(config[names[k][0]]) as fh:
                    result[k] = fh.read()
    return result


_call_counter = [0]


def _next_call():
    _call_counter[0] += 1
    return _call_counter[0]
