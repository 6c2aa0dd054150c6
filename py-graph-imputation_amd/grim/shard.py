"""
Multi-GPU driver: one process per GPU, contiguous shards of the subject file, graph replicated
per GPU, NO collective on the data path; rank 0 concatenates the per-rank outputs in rank order.

This is the split the reference's scripts/runfile_mp.py:109-148 intends (`split -l ceil(N/G)` +
one worker per chunk), with two differences: workers are GPU ranks, and `.miss/.problem` keep the
GLOBAL line index (the reference's per-chunk runs restart at 0).

Launch: torchrun --nproc-per-node N --master-addr 127.0.0.1 your_script.py  ->  impute_sharded(conf)
"""

import os

OUTPUT_KEYS = ("umug", "umug_pops", "pmug", "pmug_pops", "miss", "problem")


def shard_range(n, rank, world):
    """[begin, end) of rank's contiguous block of ceil(n/world) lines."""
    per = -(-n // world) if world > 0 else n
    return min(n, rank * per), min(n, (rank + 1) * per)


def merge_texts(per_rank):
    """per_rank: list (rank order) of dicts of output texts -> one dict, rank order preserved."""
    return {k: "".join(t.get(k, "") for t in per_rank) for k in OUTPUT_KEYS}


def impute_sharded(conf_file, hap_pop_pair=False, graph=None, compute=None, project_dir_graph="",
                   project_dir_in_file=""):
    """Run `impute` across the ranks of the current torch.distributed job (or alone if there is
    none).  `compute(config, lines, line_offset) -> texts` can be injected (tests); the default
    runs the HIP engine on this rank's GPU.  Returns the merged texts on rank 0, None elsewhere."""
    import pathlib

    from .run_impute_def import load_config

    rank, world, dist = 0, 1, None
    try:
        import torch.distributed as dist_mod

        if dist_mod.is_available() and dist_mod.is_initialized():
            dist = dist_mod
            rank, world = dist.get_rank(), dist.get_world_size()
    except ImportError:
        pass

    config, out_dir = load_config(conf_file, project_dir_graph, project_dir_in_file)
    with open(config["imputation_input_file"]) as fh:
        lines = fh.readlines()
    lo, hi = shard_range(len(lines), rank, world)

    if compute is None:
        from .imputation.impute import Imputation
        from .imputation.networkx_graph import Graph

        if graph is None:
            graph = Graph(config).build_graph(config["node_file"], config["top_links_file"], config["edges_file"])
        imp = Imputation(graph, config, device=int(os.environ.get("LOCAL_RANK", rank)))

        def compute(cfg, shard, offset):
            return imp.impute_lines(shard, cfg, em_mr=hap_pop_pair, line_offset=offset)

    mine = compute(config, lines[lo:hi], lo)
    if dist is None:
        gathered = [mine]
    else:
        gathered = [None] * world if rank == 0 else None
        dist.gather_object(mine, gathered, dst=0)
    if rank != 0:
        return None
    merged = merge_texts(gathered)
    pathlib.Path(out_dir).mkdir(parents=False, exist_ok=True)
    from .imputation.impute import Imputation as _I

    _I.write_outputs(config, merged)
    return merged
