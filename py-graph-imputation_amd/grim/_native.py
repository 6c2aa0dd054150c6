"""
ctypes binding of libgrim_hip.so (include/grim_hip.h).  There is no CPU fallback: if the
library is missing or no HIP device is visible, every compute entry point raises.
"""

import ctypes as C
import os

import numpy as np

MAXL = 5
ABITS = 12
MAXPH = 16
MAXPOP = 64
TOPCAP = 128
MAXLADDER = 64
MAXROWS = 8

ST_OK, ST_MISS, ST_UNSUPPORTED, ST_NOPHASE = 0, 1, 2, 3
T_UMUG, T_UMUG_POPS, T_PMUG, T_PMUG_POPS = 0, 1, 2, 3

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GRIM_LIB") or os.path.join(os.path.dirname(_HERE), "libgrim_hip.so")


class GraphDesc(C.Structure):
    _fields_ = [
        ("n_nodes", C.c_uint32), ("n_pops", C.c_uint32), ("n_loci", C.c_uint32), ("full_mask", C.c_uint32),
        ("node_key", C.c_void_p), ("node_mask", C.c_void_p), ("freq", C.c_void_p),
        ("a_start", C.c_void_p), ("a_nbr", C.c_void_p), ("n_a_nbr", C.c_uint64),
        ("b_conn", C.c_void_p), ("b_start", C.c_void_p), ("b_nbr", C.c_void_p),
        ("n_conn", C.c_uint32), ("n_b_nbr", C.c_uint64),
        ("lab_start", C.c_void_p), ("lab_nodes", C.c_void_p), ("label_order_bad", C.c_uint32), ("reserved", C.c_uint32),
    ]


class Params(C.Structure):
    _fields_ = [
        ("ladder", C.c_double * MAXLADDER), ("n_ladder", C.c_int32), ("top_n", C.c_uint32),
        ("opt_threshold", C.c_uint64), ("n_results", C.c_uint32), ("n_pop_results", C.c_uint32),
        ("out_muug", C.c_uint8), ("out_haps", C.c_uint8), ("planb", C.c_uint8), ("em_mr", C.c_uint8), ("em", C.c_uint8),
        ("save_mode", C.c_uint8), ("eps_nonpositive", C.c_uint8),
        ("pop_rank", C.c_uint8 * MAXPOP), ("factor_missing_pow", C.c_double * (MAXL + 1)),
        ("planb_rows", C.c_uint8), ("planb_nblk", C.c_uint8 * MAXROWS),
        ("planb_blk", (C.c_uint8 * MAXL) * MAXROWS),
    ]


class BatchDesc(C.Structure):
    _fields_ = [
        ("n_subjects", C.c_uint32), ("subjects", C.c_void_p), ("tokens", C.c_void_p), ("n_tokens", C.c_uint64),
        ("priors", C.c_void_p), ("n_priors", C.c_uint32),
    ]


# numpy mirrors of the plain-data structs
SUBJECT_DT = np.dtype([
    ("tok_off", "<u4"), ("prior_idx", "<u2"), ("n_loci", "u1"), ("flags", "u1"),
    ("slot", "u1", (MAXL,)), ("pad", "u1", (3,)),
    ("cnt", "<u2", (MAXL, 2)), ("wid", "<u2", (MAXL, 2)), ("reserved", "<u4", (2,)),
], align=False)
assert SUBJECT_DT.itemsize == 64

RESULT_DT = np.dtype([
    ("status", "u1"), ("plan", "u1"), ("reason", "u1"), ("plan_phased", "u1"),
    ("n_pairs", "<u4"), ("n_genotypes", "<u4"),
    ("row_off", "<u4", (4,)), ("n_rows", "<u4", (4,)), ("pad2", "<u4"), ("max_prob", "<f8"),
], align=False)
assert RESULT_DT.itemsize == 56

ROW_DT = np.dtype([("a", "<u8"), ("b", "<u8"), ("prob", "<f8"), ("popa", "<u4"), ("popb", "<u4")])
assert ROW_DT.itemsize == 32

_lib = None


class NativeError(RuntimeError):
    pass


def _share_hip_runtime_with_torch():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so.  Two HIP runtimes in one process do not
    work (whoever comes second sees no GPU), and torch.distributed's RCCL backend needs torch's, so
    when a torch wheel is installed its runtime is loaded first and this library binds to it.
    GRIM_HIP_RUNTIME=system keeps the system runtime (then do not use torch.cuda in the process)."""
    if os.environ.get("GRIM_HIP_RUNTIME") == "system":
        return
    try:
        import importlib.util

        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:
        pass


def lib():
    """Load the shared library (once).  Raises NativeError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeError(
            "libgrim_hip.so not found at %s -- build it with `python __graft_entry__.py` "
            "(hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
    _share_hip_runtime_with_torch()
    L = C.CDLL(LIB_PATH)
    L.grim_create.restype = C.c_void_p
    L.grim_create.argtypes = [C.c_int]
    L.grim_export_engine.restype = C.c_int
    L.grim_export_engine.argtypes = [C.c_void_p]
    L.grim_destroy.argtypes = [C.c_void_p]
    L.grim_last_error.restype = C.c_char_p
    L.grim_last_error.argtypes = [C.c_void_p]
    L.grim_device_count.restype = C.c_int
    L.grim_graph_upload.restype = C.c_void_p
    L.grim_graph_upload.argtypes = [C.c_void_p, C.POINTER(GraphDesc)]
    L.grim_graph_free.argtypes = [C.c_void_p]
    L.grim_graph_device_bytes.restype = C.c_uint64
    L.grim_graph_device_bytes.argtypes = [C.c_void_p]
    L.grim_batch_upload.restype = C.c_void_p
    L.grim_batch_upload.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Params), C.POINTER(BatchDesc)]
    L.grim_batch_run.restype = C.c_int
    L.grim_batch_run.argtypes = [C.c_void_p]
    L.grim_batch_kernel_ms.restype = C.c_double
    L.grim_batch_kernel_ms.argtypes = [C.c_void_p, C.c_int]
    L.grim_batch_counters.restype = C.c_int
    L.grim_batch_counters.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    L.grim_batch_total_rows.restype = C.c_uint32
    L.grim_batch_total_rows.argtypes = [C.c_void_p]
    L.grim_batch_results.restype = C.c_int
    L.grim_batch_results.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.grim_batch_free.argtypes = [C.c_void_p]
    L.grim_batch_run_repeat.restype = C.c_int
    L.grim_batch_run_repeat.argtypes = [C.c_void_p, C.c_uint32]
    L.grim_batch_set_timing.restype = C.c_int
    L.grim_batch_set_timing.argtypes = [C.c_void_p, C.c_int]
    _lib = L
    return L


EXPORTS = [
    "grim_create", "grim_destroy", "grim_last_error", "grim_device_count", "grim_export_engine", "grim_graph_upload", "grim_graph_free",
    "grim_graph_device_bytes", "grim_batch_upload", "grim_batch_run", "grim_batch_kernel_ms", "grim_batch_counters",
    "grim_batch_total_rows", "grim_batch_results", "grim_batch_free", "grim_batch_set_timing", "grim_batch_run_repeat",
]


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Context:
    """One per GPU (grim_ctx)."""

    def __init__(self, device=0):
        L = lib()
        self.device = device
        self.h = L.grim_create(device)
        if not self.h:
            raise NativeError("grim_create(%d) failed: %s" % (device, L.grim_last_error(None).decode()))

    def error(self):
        return lib().grim_last_error(self.h).decode()

    def export_engine(self):
        """SDMA engine bit of the result downloads (> 1), 0 = copy kernel, -1 = hipMemcpyAsync (grim_export_engine)"""
        return int(lib().grim_export_engine(self.h))

    def close(self):
        if self.h:
            lib().grim_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_contexts = {}


def default_context(device=None):
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0")) if os.environ.get("GRIM_DEVICE") is None else int(os.environ["GRIM_DEVICE"])
        n = lib().grim_device_count()
        if n > 0:
            device %= n
    if device not in _contexts:
        _contexts[device] = Context(device)
    return _contexts[device]


class DeviceGraph:
    """Device-resident graph (grim_graph)."""

    def __init__(self, ctx, arrays):
        L = lib()
        self.ctx = ctx
        d = GraphDesc()
        self._keep = arrays  # host arrays must outlive the upload call only, kept for debugging
        d.n_nodes = arrays["n_nodes"]
        d.n_pops = arrays["n_pops"]
        d.n_loci = arrays["n_loci"]
        d.full_mask = arrays["full_mask"]
        for k in ("node_key", "node_mask", "freq", "a_start", "a_nbr", "b_conn", "b_start", "b_nbr", "lab_start", "lab_nodes"):
            setattr(d, k, _ptr(arrays[k]))
        d.n_a_nbr = arrays["a_nbr"].shape[0]
        d.n_conn = arrays["b_start"].shape[0] - 1
        d.n_b_nbr = arrays["b_nbr"].shape[0]
        d.label_order_bad = int(arrays.get("label_order_bad", 0))
        self.h = L.grim_graph_upload(ctx.h, C.byref(d))
        if not self.h:
            raise NativeError("grim_graph_upload failed: " + ctx.error())

    def device_bytes(self):
        return int(lib().grim_graph_device_bytes(self.h))

    def close(self):
        if self.h:
            lib().grim_graph_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceBatch:
    """Subjects resident in HBM + result/scratch buffers (grim_batch)."""

    def __init__(self, ctx, dgraph, params, subjects, tokens, priors):
        L = lib()
        self.ctx = ctx
        self.n = int(subjects.shape[0])
        d = BatchDesc()
        tokens = np.ascontiguousarray(tokens, dtype=np.uint16)
        priors = np.ascontiguousarray(priors, dtype=np.float64)
        subjects = np.ascontiguousarray(subjects)
        d.n_subjects = self.n
        d.subjects = _ptr(subjects)
        d.tokens = _ptr(tokens)
        d.n_tokens = tokens.shape[0]
        d.priors = _ptr(priors)
        d.n_priors = priors.shape[0]
        self._keep = (subjects, tokens, priors)
        self.h = L.grim_batch_upload(ctx.h, dgraph.h, C.byref(params), C.byref(d))
        if not self.h:
            raise NativeError("grim_batch_upload failed: " + ctx.error())

    def run(self):
        rc = lib().grim_batch_run(self.h)
        if rc != 0:
            raise NativeError("grim_batch_run failed (%d): %s" % (rc, self.ctx.error()))

    def run_repeat(self, n):
        rc = lib().grim_batch_run_repeat(self.h, int(n))
        if rc != 0:
            raise NativeError("grim_batch_run failed (%d): %s" % (rc, self.ctx.error()))

    def set_timing(self, on=True):
        """start/stop hipEvents around every kernel of a run (resets the accumulated means); see grim_batch_set_timing"""
        lib().grim_batch_set_timing(self.h, 1 if on else 0)

    def kernel_ms(self, which=0):
        return float(lib().grim_batch_kernel_ms(self.h, which))

    def counters(self):
        out = (C.c_uint64 * 4)()
        lib().grim_batch_counters(self.h, out)
        return [int(x) for x in out]

    def results(self):
        L = lib()
        nrows = int(L.grim_batch_total_rows(self.h))
        res = np.zeros(self.n, dtype=RESULT_DT)
        rows = np.zeros(max(nrows, 1), dtype=ROW_DT)
        rc = L.grim_batch_results(self.h, _ptr(res), _ptr(rows))
        if rc != 0:
            raise NativeError("grim_batch_results failed: " + self.ctx.error())
        return res, rows[:nrows]

    def close(self):
        if self.h:
            lib().grim_batch_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ======================================================================================================
# host-side helpers of the library (C++: allele dictionary, tokenizer, formatter) -- no GPU needed
# ======================================================================================================
K_DEVICE, K_PROBLEM_ID, K_PROBLEM_RAW, K_MISS_NO_DEVICE, K_UNSUPPORTED, K_UNSUPPORTED_GL = 0, 1, 2, 3, 4, 5

EXPORTS += [
    "grim_dict_create", "grim_dict_free", "grim_dict_set_locus", "grim_dict_intern", "grim_dict_find", "grim_dict_name",
    "grim_dict_count", "grim_tokenize", "grim_parsed_free", "grim_parsed_lines", "grim_parsed_subjects",
    "grim_parsed_subject_array", "grim_parsed_tokens", "grim_parsed_kinds", "grim_parsed_dev_index",
    "grim_parsed_n_races", "grim_parsed_race", "grim_parsed_id", "grim_parsed_set_kind", "grim_parsed_set_flags", "grim_format", "grim_text_get", "grim_text_free",
    "grim_format_double", "grim_hostgraph_load_csv", "grim_hostgraph_desc", "grim_hostgraph_free", "grim_graphgen_csv", "grim_hostgraph_from_hpf",
    "grim_parsed_allele", "grim_prior_matrix", "grim_stream_open", "grim_stream_write", "grim_stream_write_borrowed", "grim_stream_write_file", "grim_stream_finish",
    "grim_stream_error", "grim_stream_text", "grim_stream_get_stats", "grim_stream_n_unsupported", "grim_stream_unsupported",
    "grim_stream_next_records", "grim_stream_release_records", "grim_stream_free", "grim_stream_write_text",
    "grim_stream_segment", "grim_stream_n_segments", "grim_stream_segment_end", "grim_stream_segment_wait", "grim_stream_segment_place", "grim_default_threads", "grim_chunk_offsets", "grim_free",
]


class PriorSpec(C.Structure):
    _fields_ = [("alpha", C.c_double), ("eta", C.c_double), ("beta", C.c_double), ("gamma", C.c_double), ("delta", C.c_double),
                ("unk_mr", C.c_uint8), ("count_by_prob", C.POINTER(C.c_double))]


class StreamOpts(C.Structure):
    _fields_ = [("chunk_lines", C.c_uint32), ("depth", C.c_uint32), ("n_threads", C.c_int32), ("line_offset", C.c_uint64),
                ("rows_per_chunk", C.c_uint64), ("want_text", C.c_uint8), ("want_log", C.c_uint8), ("want_records", C.c_uint8),
                ("timing", C.c_uint8), ("rows_exact", C.c_uint8), ("placed", C.c_uint8), ("out_path", C.c_char_p * 6), ("mask_ids", C.c_char_p), ("mask_fixed", C.c_void_p),
                ("n_masks", C.c_uint32)]


class StreamStats(C.Structure):
    _fields_ = [("lines", C.c_uint64), ("subjects", C.c_uint64), ("chunks", C.c_uint64), ("reruns", C.c_uint64),
                ("unsupported", C.c_uint64), ("wall_s", C.c_double), ("tokenize_cpu_s", C.c_double), ("format_cpu_s", C.c_double),
                ("write_cpu_s", C.c_double), ("device_s", C.c_double), ("kernel_ms", C.c_double * 7), ("counters", C.c_uint64 * 4),
                ("text_bytes", C.c_uint64 * 7), ("bytes_h2d", C.c_uint64), ("bytes_d2h", C.c_uint64)]


class StreamRecords(C.Structure):
    _fields_ = [("first_line", C.c_uint64), ("n_lines", C.c_uint32), ("kinds", C.c_void_p), ("res", C.c_void_p),
                ("rows", C.c_void_p), ("chunk", C.c_void_p)]


_host_ready = False


def host_lib():
    global _host_ready
    L = lib()
    if _host_ready:
        return L
    L.grim_dict_create.restype = C.c_void_p
    L.grim_dict_create.argtypes = [C.c_uint32]
    L.grim_dict_free.argtypes = [C.c_void_p]
    L.grim_dict_set_locus.argtypes = [C.c_void_p, C.c_uint32, C.c_char_p]
    L.grim_dict_intern.restype = C.c_int32
    L.grim_dict_intern.argtypes = [C.c_void_p, C.c_uint32, C.c_char_p]
    L.grim_dict_find.restype = C.c_int32
    L.grim_dict_find.argtypes = [C.c_void_p, C.c_uint32, C.c_char_p]
    L.grim_dict_name.restype = C.c_char_p
    L.grim_dict_name.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
    L.grim_dict_count.restype = C.c_uint32
    L.grim_dict_count.argtypes = [C.c_void_p, C.c_uint32]
    L.grim_tokenize.restype = C.c_void_p
    L.grim_tokenize.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, C.c_int, C.c_int]
    L.grim_parsed_free.argtypes = [C.c_void_p]
    L.grim_parsed_lines.restype = C.c_uint32
    L.grim_parsed_lines.argtypes = [C.c_void_p]
    L.grim_parsed_subjects.restype = C.c_uint32
    L.grim_parsed_subjects.argtypes = [C.c_void_p]
    L.grim_parsed_subject_array.restype = C.c_void_p
    L.grim_parsed_subject_array.argtypes = [C.c_void_p]
    L.grim_parsed_tokens.restype = C.c_void_p
    L.grim_parsed_tokens.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    L.grim_parsed_kinds.restype = C.c_void_p
    L.grim_parsed_kinds.argtypes = [C.c_void_p]
    L.grim_parsed_dev_index.restype = C.c_void_p
    L.grim_parsed_dev_index.argtypes = [C.c_void_p]
    L.grim_parsed_n_races.restype = C.c_uint32
    L.grim_parsed_n_races.argtypes = [C.c_void_p]
    L.grim_parsed_race.restype = C.c_char_p
    L.grim_parsed_race.argtypes = [C.c_void_p, C.c_uint32, C.c_int]
    L.grim_parsed_id.restype = C.c_void_p
    L.grim_parsed_id.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
    L.grim_parsed_set_kind.argtypes = [C.c_void_p, C.c_uint32, C.c_uint8]
    L.grim_parsed_set_flags.argtypes = [C.c_void_p, C.c_uint32, C.c_uint8]
    L.grim_format.restype = C.c_void_p
    L.grim_format.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Params), C.POINTER(C.c_char_p), C.c_uint32, C.c_void_p,
                              C.c_void_p, C.c_uint64, C.c_void_p, C.c_int]
    L.grim_text_get.restype = C.c_void_p
    L.grim_text_get.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint64)]
    L.grim_text_free.argtypes = [C.c_void_p]
    L.grim_format_double.restype = C.c_int
    L.grim_format_double.argtypes = [C.c_double, C.c_char_p, C.c_int]
    L.grim_hostgraph_load_csv.restype = C.c_void_p
    L.grim_hostgraph_load_csv.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_uint64]
    L.grim_hostgraph_desc.restype = C.c_int
    L.grim_hostgraph_desc.argtypes = [C.c_void_p, C.POINTER(GraphDesc)]
    L.grim_hostgraph_free.argtypes = [C.c_void_p]
    L.grim_graphgen_csv.restype = C.c_int
    L.grim_graphgen_csv.argtypes = [C.c_char_p, C.POINTER(C.c_char_p), C.POINTER(C.c_double), C.c_uint32, C.POINTER(C.c_char_p),
                                    C.POINTER(C.c_uint32), C.c_uint32, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p,
                                    C.c_char_p, C.c_uint64]
    L.grim_hostgraph_from_hpf.restype = C.c_void_p
    L.grim_hostgraph_from_hpf.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_char_p), C.POINTER(C.c_double), C.c_uint32,
                                          C.POINTER(C.c_char_p), C.POINTER(C.c_uint32), C.c_uint32, C.c_char_p, C.c_char_p, C.c_char_p,
                                          C.c_char_p, C.c_char_p, C.c_uint64]
    L.grim_parsed_allele.restype = C.c_void_p
    L.grim_parsed_allele.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]
    L.grim_prior_matrix.restype = C.c_int
    L.grim_prior_matrix.argtypes = [C.POINTER(PriorSpec), C.POINTER(C.c_char_p), C.c_uint32, C.c_char_p, C.c_char_p, C.c_void_p]
    L.grim_stream_open.restype = C.c_void_p
    L.grim_stream_open.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Params), C.POINTER(PriorSpec), C.POINTER(C.c_char_p),
                                   C.c_uint32, C.POINTER(StreamOpts)]
    L.grim_stream_write.restype = C.c_int
    L.grim_stream_write.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64]
    L.grim_stream_write_borrowed.restype = C.c_int
    L.grim_stream_write_borrowed.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64]
    L.grim_stream_write_file.restype = C.c_int
    L.grim_stream_write_file.argtypes = [C.c_void_p, C.c_char_p]
    L.grim_stream_finish.restype = C.c_int
    L.grim_stream_finish.argtypes = [C.c_void_p]
    L.grim_stream_error.restype = C.c_char_p
    L.grim_stream_error.argtypes = [C.c_void_p]
    L.grim_stream_text.restype = C.c_void_p
    L.grim_stream_text.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint64)]
    L.grim_stream_get_stats.restype = C.c_int
    L.grim_stream_get_stats.argtypes = [C.c_void_p, C.POINTER(StreamStats)]
    L.grim_stream_n_unsupported.restype = C.c_uint64
    L.grim_stream_n_unsupported.argtypes = [C.c_void_p]
    L.grim_stream_unsupported.restype = C.c_int
    L.grim_stream_unsupported.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.c_void_p),
                                          C.POINTER(C.c_uint32)]
    L.grim_stream_next_records.restype = C.c_int
    L.grim_stream_next_records.argtypes = [C.c_void_p, C.POINTER(StreamRecords)]
    L.grim_stream_release_records.restype = C.c_int
    L.grim_stream_release_records.argtypes = [C.c_void_p, C.POINTER(StreamRecords)]
    L.grim_stream_free.argtypes = [C.c_void_p]
    L.grim_stream_write_text.restype = C.c_int
    L.grim_stream_write_text.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64]
    L.grim_stream_segment.restype = C.c_int
    L.grim_stream_segment.argtypes = [C.c_void_p, C.c_uint64]
    L.grim_stream_n_segments.restype = C.c_uint32
    L.grim_stream_n_segments.argtypes = [C.c_void_p]
    L.grim_stream_segment_end.restype = C.c_int
    L.grim_stream_segment_end.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint64)]
    L.grim_stream_segment_wait.restype = C.c_int
    L.grim_stream_segment_wait.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint64)]
    L.grim_stream_segment_place.restype = C.c_int
    L.grim_stream_segment_place.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint64)]
    L.grim_default_threads.restype = C.c_uint32
    L.grim_chunk_offsets.restype = C.c_int64
    L.grim_chunk_offsets.argtypes = [C.c_char_p, C.c_uint32, C.POINTER(C.POINTER(C.c_uint64))]
    L.grim_free.argtypes = [C.c_void_p]
    _host_ready = True
    return L


def host_threads():
    n = int(os.environ.get("GRIM_HOST_THREADS", "0"))
    if n <= 0:
        n = min(16, os.cpu_count() or 1)
    return n


class AlleleDict:
    """grim_dict: the one allele dictionary shared by the graph loader, the tokenizer and the formatter."""

    def __init__(self, slot_locus):
        L = host_lib()
        self.n = len(slot_locus)
        self.h = L.grim_dict_create(self.n)
        if not self.h:
            raise NativeError("grim_dict_create failed")
        for s, name in enumerate(slot_locus):
            L.grim_dict_set_locus(self.h, s, name.encode())
        self._cache = [dict() for _ in slot_locus]

    def intern(self, slot, allele):
        i = self._cache[slot].get(allele)
        if i is None:
            i = host_lib().grim_dict_intern(self.h, slot, allele.encode())
            if i < 0:
                raise OverflowError("more than %d alleles at locus slot %d" % ((1 << ABITS) - 2, slot))
            self._cache[slot][allele] = i
        return i

    def name(self, slot, idx):
        s = host_lib().grim_dict_name(self.h, slot, idx)
        return None if s is None else s.decode()

    def count(self, slot):
        return int(host_lib().grim_dict_count(self.h, slot))

    def __del__(self):
        try:
            if self.h:
                host_lib().grim_dict_free(self.h)
                self.h = None
        except Exception:
            pass


class Parsed:
    """grim_parsed: a block of input lines tokenised by the library."""

    def __init__(self, adict, text_bytes, planb):
        L = host_lib()
        self.h = L.grim_tokenize(adict.h, text_bytes, len(text_bytes), 1 if planb else 0, host_threads())
        if not self.h:
            raise NativeError("grim_tokenize failed")
        self.n_lines = int(L.grim_parsed_lines(self.h))
        self.n_subjects = int(L.grim_parsed_subjects(self.h))

    def _array(self, ptr, dtype, n):
        if n == 0:
            return np.zeros(0, dtype=dtype)
        buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
        return np.frombuffer(buf, dtype=dtype, count=n).copy()

    def subjects(self):
        return self._array(host_lib().grim_parsed_subject_array(self.h), SUBJECT_DT, self.n_subjects)

    def tokens(self):
        n = C.c_uint64(0)
        ptr = host_lib().grim_parsed_tokens(self.h, C.byref(n))
        return self._array(ptr, np.uint16, int(n.value))

    def kinds(self):
        return self._array(host_lib().grim_parsed_kinds(self.h), np.uint8, self.n_lines)

    def dev_index(self):
        return self._array(host_lib().grim_parsed_dev_index(self.h), np.int32, self.n_lines)

    def races(self):
        L = host_lib()
        return [(L.grim_parsed_race(self.h, i, 0).decode(), L.grim_parsed_race(self.h, i, 1).decode())
                for i in range(int(L.grim_parsed_n_races(self.h)))]

    def set_kind(self, line, kind):
        host_lib().grim_parsed_set_kind(self.h, line, kind)

    def set_flags(self, line, flags):
        host_lib().grim_parsed_set_flags(self.h, line, flags)

    def subject_id(self, line):
        n = C.c_uint32(0)
        ptr = host_lib().grim_parsed_id(self.h, line, C.byref(n))
        return C.string_at(ptr, n.value).decode() if ptr else None

    def allele(self, line, slot, idx):
        """text of allele `idx` at locus slot `slot` as line `line` uses it (dictionary allele or the line's own)"""
        n = C.c_uint32(0)
        ptr = host_lib().grim_parsed_allele(self.h, line, slot, idx, C.byref(n))
        return C.string_at(ptr, n.value).decode() if ptr else None

    def format(self, adict, params, pops, res, rows, line_offset=0, skip=None, as_bytes=False):
        L = host_lib()
        names = (C.c_char_p * len(pops))(*[p.encode() for p in pops])
        res = np.ascontiguousarray(res)
        rows = np.ascontiguousarray(rows)
        skip_p = None
        if skip is not None:
            skip = np.ascontiguousarray(skip, dtype=np.uint8)
            skip_p = _ptr(skip)
        t = L.grim_format(adict.h, self.h, C.byref(params), names, len(pops), _ptr(res) if res.size else None,
                          _ptr(rows) if rows.size else None, line_offset, skip_p, host_threads())
        if not t:
            raise NativeError("grim_format failed")
        out = {}
        for k, key in enumerate(("umug", "umug_pops", "pmug", "pmug_pops", "miss", "problem")):
            n = C.c_uint64(0)
            ptr = L.grim_text_get(t, k, C.byref(n))
            if as_bytes:
                out[key] = C.string_at(ptr, n.value) if n.value else b""
            else:
                out[key] = C.string_at(ptr, n.value).decode() if n.value else ""
        L.grim_text_free(t)
        return out

    def close(self):
        if self.h:
            host_lib().grim_parsed_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def chunk_offsets(path, chunk_lines):
    """grim_chunk_offsets: byte offsets of every chunk_lines-th line start of the file, the file size last"""
    L = host_lib()
    out = C.POINTER(C.c_uint64)()
    n = L.grim_chunk_offsets(os.fsencode(path), int(chunk_lines), C.byref(out))
    if n < 0:
        raise OSError("grim_chunk_offsets(%s) failed" % path)
    try:
        return [int(out[i]) for i in range(n)]
    finally:
        L.grim_free(out)


def format_double(x):
    buf = C.create_string_buffer(48)
    n = host_lib().grim_format_double(float(x), buf, 48)
    return buf.value[:n].decode()


def load_graph_csv(adict, full_loci, nodes_csv, top_links_csv, edges_csv):
    """grim_hostgraph_load_csv: the three graph CSVs -> dict of numpy arrays (same keys as the Python loader
    builds; the arrays are copies, the C object is freed before returning)."""
    L = host_lib()
    err = C.create_string_buffer(512)
    h = L.grim_hostgraph_load_csv(adict.h, full_loci.encode(), os.fsencode(nodes_csv), os.fsencode(top_links_csv),
                                  os.fsencode(edges_csv), err, len(err))
    return _hostgraph_arrays(h, err)


def _hostgraph_arrays(h, err):
    """a grim_hostgraph handle -> dict of numpy arrays (copies); frees the handle"""
    L = host_lib()
    if not h:
        msg = err.value.decode()
        if msg.startswith("graph: the highest-numbered vertex"):
            raise IndexError(msg)
        raise ValueError(msg)
    try:
        d = GraphDesc()
        L.grim_hostgraph_desc(h, C.byref(d))

        def arr(ptr, dtype, n):
            if n == 0:
                return np.zeros(0, dtype=dtype)
            buf = (C.c_char * (int(n) * np.dtype(dtype).itemsize)).from_address(ptr)
            return np.frombuffer(buf, dtype=dtype, count=int(n)).copy()

        V, P = int(d.n_nodes), int(d.n_pops)
        return {
            "n_nodes": V, "n_pops": P, "n_loci": int(d.n_loci), "full_mask": int(d.full_mask),
            "node_key": arr(d.node_key, np.uint64, V), "node_mask": arr(d.node_mask, np.uint8, V),
            "freq": arr(d.freq, np.float64, V * P).reshape(V, P),
            "a_start": arr(d.a_start, np.uint32, V + 1), "a_nbr": arr(d.a_nbr, np.uint32, d.n_a_nbr),
            "b_conn": arr(d.b_conn, np.uint32, V * MAXL), "b_start": arr(d.b_start, np.uint32, int(d.n_conn) + 1),
            "b_nbr": arr(d.b_nbr, np.uint32, d.n_b_nbr),
            "lab_start": arr(d.lab_start, np.uint32, (1 << MAXL) + 1), "lab_nodes": arr(d.lab_nodes, np.uint32, V),
        }
    finally:
        L.grim_hostgraph_free(h)


def graphgen_csv(hpf_csv, pops, cutoffs, loci_map, nodes_csv, edges_csv, top_links_csv, info_csv):
    """grim_graphgen_csv: hpf.csv -> nodes.csv, edges.csv, top_links.csv, info_node.csv."""
    L = host_lib()
    err = C.create_string_buffer(512)
    pop_arr = (C.c_char_p * len(pops))(*[p.encode() for p in pops])
    cut_arr = (C.c_double * len(pops))(*[float(c) for c in cutoffs])
    names = list(loci_map.keys())
    name_arr = (C.c_char_p * len(names))(*[n.encode() for n in names])
    idx_arr = (C.c_uint32 * len(names))(*[int(loci_map[n]) for n in names])
    rc = L.grim_graphgen_csv(os.fsencode(hpf_csv), pop_arr, cut_arr, len(pops), name_arr, idx_arr, len(names),
                             os.fsencode(nodes_csv), os.fsencode(edges_csv), os.fsencode(top_links_csv), os.fsencode(info_csv),
                             err, len(err))
    if rc != 0:
        raise ValueError(err.value.decode())


def graph_from_hpf(adict, full_loci, hpf_csv, pops, cutoffs, loci_map, csv_paths=None):
    """grim_hostgraph_from_hpf: hpf.csv -> the loader's arrays, generator and loader back to back in memory.
    csv_paths: optional (nodes, edges, top_links, info_node) paths to ALSO write the four CSVs (None entries are skipped)."""
    L = host_lib()
    err = C.create_string_buffer(512)
    pop_arr = (C.c_char_p * len(pops))(*[p.encode() for p in pops])
    cut_arr = (C.c_double * len(pops))(*[float(c) for c in cutoffs])
    names = list(loci_map.keys())
    name_arr = (C.c_char_p * len(names))(*[n.encode() for n in names])
    idx_arr = (C.c_uint32 * len(names))(*[int(loci_map[n]) for n in names])
    paths = [os.fsencode(p) if p else None for p in (csv_paths or (None, None, None, None))]
    h = L.grim_hostgraph_from_hpf(adict.h, full_loci.encode(), os.fsencode(hpf_csv), pop_arr, cut_arr, len(pops), name_arr, idx_arr,
                                  len(names), paths[0], paths[1], paths[2], paths[3], err, len(err))
    return _hostgraph_arrays(h, err)


def prior_spec(priority, unk_priors, count_by_prob=None):
    """grim_prior_spec from the conf's "priority" dict, UNK_priors and the population counts (kept alive by the caller)"""
    ps = PriorSpec()
    ps.alpha, ps.eta, ps.beta, ps.gamma, ps.delta = (float(priority[k]) for k in ("alpha", "eta", "beta", "gamma", "delta"))
    ps.unk_mr = 1 if unk_priors == "MR" else 0
    keep = None
    if count_by_prob is not None:
        keep = np.ascontiguousarray(count_by_prob, dtype=np.float64)
        ps.count_by_prob = keep.ctypes.data_as(C.POINTER(C.c_double))
    return ps, keep


def prior_matrix(ps, pops, race1, race2):
    """calc_priority_matrix in the library (grim_prior_matrix) -> P x P float64"""
    names = (C.c_char_p * len(pops))(*[p.encode() for p in pops])
    out = np.zeros((len(pops), len(pops)), dtype=np.float64)
    rc = host_lib().grim_prior_matrix(C.byref(ps), names, len(pops), (race1 or "").encode(), (race2 or "").encode(), _ptr(out))
    if rc != 0:
        raise NativeError("grim_prior_matrix failed")
    return out


TEXT_KEYS = ("umug", "umug_pops", "pmug", "pmug_pops", "miss", "problem")


class Stream:
    """grim_stream: the chunked tokenizer -> device -> formatter pipeline (impute_file's loop)."""

    def __init__(self, ctx, dgraph, adict, params, ps, pops, out_paths=None, want_text=True, want_log=False, want_records=False,
                 chunk_lines=0, depth=0, n_threads=0, line_offset=0, rows_per_chunk=0, timing=False, masks=None, rows_exact=False,
                 placed=False):
        L = host_lib()
        self.ctx = ctx
        o = StreamOpts()
        o.chunk_lines = int(chunk_lines or int(os.environ.get("GRIM_CHUNK_LINES", "0")))
        o.depth = int(depth or int(os.environ.get("GRIM_STREAM_DEPTH", "0")))
        o.n_threads = int(n_threads or int(os.environ.get("GRIM_HOST_THREADS", "0")))
        o.line_offset = int(line_offset)
        o.rows_per_chunk = int(rows_per_chunk or int(os.environ.get("GRIM_ROWS_PER_CHUNK", "0")))
        o.want_text = 1 if want_text else 0
        o.want_log = 1 if want_log else 0
        o.want_records = 1 if want_records else 0
        o.timing = 1 if timing else 0
        o.rows_exact = 1 if rows_exact else 0
        o.placed = 1 if placed else 0
        self._keep = [params, ps]
        if out_paths:
            for k, key in enumerate(TEXT_KEYS):
                if out_paths.get(key):
                    o.out_path[k] = os.fsencode(out_paths[key])
        if masks is not None:
            ids = b"".join(str(k).encode() + b"\0" for k in masks.keys())
            fixed = np.array([int(v) for v in masks.values()], dtype=np.uint8)
            self._keep += [ids, fixed]
            o.mask_ids = ids if len(masks) else b"\0"
            o.mask_fixed = _ptr(fixed) if len(masks) else None
            o.n_masks = len(masks)
        names = (C.c_char_p * len(pops))(*[p.encode() for p in pops])
        self.line_offset = int(line_offset)
        self._lent = []  # buffers lent to the stream (write(..., borrowed=True)): alive until close()
        self.h = L.grim_stream_open(ctx.h, dgraph.h, adict.h, C.byref(params), C.byref(ps), names, len(pops), C.byref(o))
        if not self.h:
            raise NativeError("grim_stream_open failed: " + ctx.error())

    def _check(self, rc):
        if rc < 0:
            raise NativeError("grim_stream: " + host_lib().grim_stream_error(self.h).decode())

    def write(self, data, borrowed=False):
        """feed bytes; `borrowed`: the stream reads them where they are (grim_stream_write_borrowed) -- this object keeps
        `data` alive until finish() / close()"""
        if borrowed:
            if not isinstance(data, bytes):
                raise TypeError("a borrowed buffer must be a bytes object")
            self._lent.append(data)
            self._check(host_lib().grim_stream_write_borrowed(self.h, data, len(data)))
        else:
            self._check(host_lib().grim_stream_write(self.h, data, len(data)))

    def write_text(self, data):
        """universal newlines, as Python's open(): "\r\n" and "\r" end a line too"""
        self._check(host_lib().grim_stream_write_text(self.h, data, len(data)))

    def segment(self, next_line_offset):
        """end the current input segment; the next byte written starts the line with this global index"""
        self._check(host_lib().grim_stream_segment(self.h, int(next_line_offset)))

    def segment_ends(self):
        """after finish(): per segment the cumulative bytes of the seven texts up to its end"""
        L = host_lib()
        out = []
        for k in range(int(L.grim_stream_n_segments(self.h))):
            v = (C.c_uint64 * 7)()
            L.grim_stream_segment_end(self.h, k, v)
            out.append([int(x) for x in v])
        return out

    def segment_wait(self, k):
        """placed output: blocks until the (closed) segment k is formatted; -> bytes of its piece of the seven texts"""
        v = (C.c_uint64 * 7)()
        self._check(host_lib().grim_stream_segment_wait(self.h, int(k), v))
        return [int(x) for x in v]

    def segment_place(self, k, base):
        """placed output: writes segment k's piece of every output file at base[t]; returns when the bytes are written"""
        v = (C.c_uint64 * 6)(*[int(x) for x in base])
        self._check(host_lib().grim_stream_segment_place(self.h, int(k), v))

    def write_file(self, path):
        self._check(host_lib().grim_stream_write_file(self.h, os.fsencode(path)))

    def finish(self):
        self._check(host_lib().grim_stream_finish(self.h))

    def text(self, which, as_bytes=False):
        n = C.c_uint64(0)
        ptr = host_lib().grim_stream_text(self.h, which, C.byref(n))
        data = C.string_at(ptr, n.value) if n.value else b""
        return data if as_bytes else data.decode()

    def stats(self):
        st = StreamStats()
        host_lib().grim_stream_get_stats(self.h, C.byref(st))
        return st

    def unsupported(self):
        L = host_lib()
        out = []
        for k in range(int(L.grim_stream_n_unsupported(self.h))):
            line, reason, ptr, n = C.c_uint64(0), C.c_uint32(0), C.c_void_p(0), C.c_uint32(0)
            L.grim_stream_unsupported(self.h, k, C.byref(line), C.byref(reason), C.byref(ptr), C.byref(n))
            out.append((int(line.value), C.string_at(ptr.value, n.value).decode() if n.value else "", int(reason.value)))
        return out

    def next_records(self):
        """records mode: -> (first_line, kinds, res, rows_base_address, handle) of the next chunk, or None at the end.
        The arrays are VIEWS of the chunk's pinned buffers: copy what you keep, then release(handle)."""
        rec = StreamRecords()
        rc = host_lib().grim_stream_next_records(self.h, C.byref(rec))
        self._check(rc)
        if rc == 0:
            return None
        n = int(rec.n_lines)
        kinds = np.frombuffer((C.c_char * n).from_address(rec.kinds), dtype=np.uint8, count=n)
        res = np.frombuffer((C.c_char * (n * RESULT_DT.itemsize)).from_address(rec.res), dtype=RESULT_DT, count=n)
        return int(rec.first_line), kinds, res, rec.rows, rec

    def release(self, rec):
        host_lib().grim_stream_release_records(self.h, C.byref(rec))

    def close(self):
        if self.h:
            host_lib().grim_stream_free(self.h)
            self.h = None
        self._lent = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
