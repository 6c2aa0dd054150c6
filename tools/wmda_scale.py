#!/usr/bin/env python3
"""
BASELINE config 5 stand-in: a synthetic WMDA-scale single-population graph (the real WMDA set needs
a network fetch) built by recombining CAU class-I / class-II blocks with Zipf-like
frequencies, and subjects of three kinds on it: fully typed, mixed (ambiguity / missing loci /
recombinants) and high ambiguity (8 alternatives per locus and side).  Reports graph size, load
and kernel times, algorithmic GB/s, and checks a sample against the CPU oracle.

    python tools/wmda_scale.py [n_haplotypes=100000] [n_subjects=20000]
"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import harness, synth
sys.path.insert(0, harness.ROOT)


def make_freqs(n_haps, seed=7):
    rows = synth.read_freqs(synth.CAU_FREQS)
    rng = np.random.default_rng(seed)
    haps = [synth.hap_alleles(h) for h, _, _ in rows]
    c1 = {}
    c2 = {}
    for h, (_, _, f) in zip(haps, rows):
        c1[(h["A"], h["B"], h["C"])] = c1.get((h["A"], h["B"], h["C"]), 0) + f
        c2[(h["DQB1"], h["DRB1"])] = c2.get((h["DQB1"], h["DRB1"]), 0) + f
    k1, k2 = list(c1), list(c2)
    p1 = np.array([c1[k] for k in k1]); p1 /= p1.sum()
    p2 = np.array([c2[k] for k in k2]); p2 /= p2.sum()
    # extra class-I blocks by swapping single alleles between existing blocks
    extra = set()
    while len(k1) + len(extra) < max(len(k1), (n_haps * 3) // max(1, len(k2))):
        a, b = k1[int(rng.integers(len(k1)))], k1[int(rng.integers(len(k1)))]
        j = int(rng.integers(3))
        nb = tuple(b[i] if i == j else a[i] for i in range(3))
        if nb not in c1:
            extra.add(nb)
    k1x = k1 + sorted(extra)
    p1x = np.concatenate([p1, np.full(len(extra), p1.min() * 0.1)]); p1x /= p1x.sum()
    seen = {}
    while len(seen) < n_haps:
        i = rng.choice(len(k1x), size=n_haps, p=p1x)
        j = rng.choice(len(k2), size=n_haps, p=p2)
        for a, b in zip(i, j):
            if len(seen) >= n_haps:
                break
            seen.setdefault((int(a), int(b)), None)
    out = []
    keys = list(seen)
    z = 1.0 / np.arange(1, len(keys) + 1) ** 0.9
    z = z / z.sum()
    rng.shuffle(z)
    for (a, b), f in zip(keys, z):
        A, B, C = k1x[a]; Q, R = k2[b]
        out.append(("%s~%s~%s~%s~%s" % (A, C, B, R, Q), "1", float("%.6g" % f)))
    return out


def main():
    n_haps = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    n_subj = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
    import __graft_entry__ as ge
    ge.build()
    os.environ["GRIM_QUIET"] = "1"
    os.environ.setdefault("GRIM_TIMING", "1")  # per-kernel events
    name = "wmda%d" % n_haps
    work = os.path.join(harness.WORK, name)
    t0 = time.time()
    rows = None
    if not os.path.exists(os.path.join(work, "output", "csv", "info_node.csv")):
        rows = make_freqs(n_haps)
        os.makedirs(os.path.join(work, "data", "freqs"), exist_ok=True)
        os.makedirs(os.path.join(work, "data", "subjects"), exist_ok=True)
        synth.write_freqs(os.path.join(work, "data", "freqs", "WMD.freqs.gz"), rows)
        conf = harness.base_conf(["WMD"])
        conf["freq_trim_threshold"] = 1e-12
        json.dump(conf, open(os.path.join(work, "graph_conf.json"), "w"))
        from graph_generation.generate_hpf import produce_hpf
        from graph_generation.generate_neo4j_multi_hpf import generate_graph
        cwd = os.getcwd(); os.chdir(work)
        produce_hpf("graph_conf.json", quiet=True); generate_graph("graph_conf.json", quiet=True)
        os.chdir(cwd)
    print("graph csv ready in %.1f s" % (time.time() - t0), flush=True)
    rows = rows or synth.read_freqs(os.path.join(work, "data", "freqs", "WMD.freqs.gz"))
    harness.POPS[name] = ["WMD"]
    conf = harness.base_conf(["WMD"])
    conf["number_of_options_threshold"] = 1000000
    gen = synth.SubjectGen(rows, 5, pops=["WMD"])
    sets = {
        "full": gen.full(n_subj),
        "mixed": gen.mixed(max(1, n_subj // 4)),
        "highamb8": gen.high_ambiguity(max(1, n_subj // 400), width=8),
    }
    from grim import _native as nat
    for kind, lines in sets.items():
        t1 = time.time()
        got, log, imp = harness.run_product(name, conf, lines, tag="w_" + kind, on_unsupported="skip")
        dt = time.time() - t1
        st = imp.last_stats
        g = imp.netGraph
        if kind == "full":
            a = g.arrays
            print("graph: %d nodes, %d top links, %d plan-B edges, host arrays %.1f MB, device %.1f MB" % (
                a["n_nodes"], len(a["a_nbr"]), len(a["b_nbr"]), g.host_bytes() / 1e6,
                g.device(nat.default_context()).device_bytes() / 1e6), flush=True)
        ctr = st["counters"]
        algo = 16 * ctr[0] + 4 * ctr[1] + 8 * ctr[2]
        print("%-9s n=%6d  wall %.2f s  device run %.4f s  kernels %.3f ms (A %.3f, B/C %.3f)  probes %d nbr %d freq %d  "
              "graph-side algorithmic %.1f MB -> %.1f GB/s  unsupported %d" % (
                  kind, len(lines), dt, st["run_s"], st["kernel_ms"], st["kernel_a_ms"], st["kernel_b_ms"], ctr[0], ctr[1], ctr[2],
                  algo / 1e6, algo / max(st["kernel_ms"], 1e-9) / 1e6, len(imp.unsupported)), flush=True)
        # parity on a sample
        m = {"full": 300, "mixed": 60, "highamb8": 2}[kind]
        t2 = time.time()
        exp, _ = harness.run_oracle(name, conf, lines[:m], tag="wo_" + kind)
        skipped = [sid for _, sid, _ in imp.unsupported]
        exp = harness.drop_subjects(exp, skipped)
        ok = True
        for k in ("umug", "pmug", "umug_pops", "pmug_pops"):
            n = len(exp[k].splitlines())
            if got[k].splitlines()[:n] != exp[k].splitlines():
                ok = False
                print("   MISMATCH in", k)
        print("   oracle sample of %d: %s (%.1f s)" % (m, "identical" if ok else "DIFFERENT", time.time() - t2), flush=True)


if __name__ == "__main__":
    main()
