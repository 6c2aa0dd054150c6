#!/usr/bin/env python3
"""
Differential fuzzing of the HIP path against the CPU oracle: random configurations x random subject
mixes (with random edge-case mutations).  Exits non-zero on the first difference and leaves the
failing case in gpurun_out/fuzz_fail.json.

    python tools/fuzz.py [rounds=20] [seed=0]
"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import harness, synth
sys.path.insert(0, harness.ROOT)
os.environ["GRIM_QUIET"] = "1"


def mutate(line, rng, by_locus):
    parts = line.split(",")
    gl = parts[1]
    r = rng.random()
    loci = gl.split("^")
    if r < 0.10:    # unknown allele replaces one side somewhere
        k = int(rng.integers(len(loci))); a, b = loci[k].split("+"); loc = a.split("*")[0]
        loci[k] = "%s*98:%02d+%s" % (loc, int(rng.integers(1, 9)), b)
    elif r < 0.15:  # both sides unknown at one locus
        k = int(rng.integers(len(loci))); loc = loci[k].split("*")[0]
        loci[k] = "%s*98:01+%s*98:02" % (loc, loc)
    elif r < 0.22:  # homozygous at one locus
        k = int(rng.integers(len(loci))); a = loci[k].split("+")[0]; loci[k] = a + "+" + a
    elif r < 0.27:  # duplicated alternative
        k = int(rng.integers(len(loci))); a, b = loci[k].split("+"); loci[k] = a + "/" + a.split("/")[0] + "+" + b
    elif r < 0.30:  # g / L suffixes
        loci = [x.replace("+", "g+", 1) if rng.random() < 0.5 else x + "L" for x in loci]
    elif r < 0.33:  # UUUU locus
        k = int(rng.integers(len(loci))); loc = loci[k].split("*")[0]; loci[k] = "%s*UUUU+%s*UUUU" % (loc, loc)
    elif r < 0.35:  # malformed
        loci[int(rng.integers(len(loci)))] = loci[0].split("+")[0]
    elif r < 0.37:
        loci.insert(int(rng.integers(len(loci) + 1)), "")
    elif r < 0.39 and len(parts) > 2:
        parts = parts[:3]
    elif r < 0.41:
        parts = parts[:2]
    elif r < 0.44:  # wide ambiguity at every locus
        loci2 = []
        for x in loci:
            a, b = x.split("+"); loc = a.split("*")[0]
            pool = by_locus[loc]
            k1 = int(rng.integers(3, 13)); k2 = int(rng.integers(1, 13))
            ext = [str(e) for e in rng.choice(pool, size=min(max(k1, k2), len(pool)), replace=False)]
            loci2.append("/".join([a] + ext[:k1]) + "+" + "/".join([b] + ext[:k2]))
        loci = loci2
    elif r < 0.48:  # a wide list of alleles the graph has never seen (Plan C with a side the label scan opens)
        k = int(rng.integers(len(loci))); a, b = loci[k].split("+"); loc = a.split("*")[0]
        loci[k] = "/".join("%s*97:%02d" % (loc, j) for j in range(1, int(rng.integers(3, 9)))) + "+" + b
    parts[1] = "^".join(loci)
    return ",".join(parts)


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    rows = synth.read_freqs(synth.CAU_FREQS)
    t0 = time.time()
    for rd in range(rounds):
        graphs = os.environ.get("GRIM_FUZZ_GRAPHS")  # e.g. "cau_bc,pop4_bc": only these (equal weights)
        if graphs:
            gname = str(rng.choice(graphs.split(",")))
        else:
            gname = str(rng.choice(["cau", "pop4", "pop9", "cau_bc", "pop4_bc"], p=[0.4, 0.35, 0.13, 0.06, 0.06]))
        pops = harness.POPS[gname]
        conf = harness.base_conf(pops)
        conf.update(harness.GRAPH_OVERRIDES.get(gname, {}))  # (a loci_map that is not alphabetical)
        conf["UNK_priors"] = "MR" if rng.random() < 0.5 else "SR"
        conf["number_of_options_threshold"] = int(rng.choice([5, 40, 300, 5000, 100000]))
        conf["max_haplotypes_number_in_phase"] = int(rng.choice([1, 3, 20, 100, 128]))
        conf["number_of_results"] = int(rng.choice([1, 3, 10, 1000]))
        conf["number_of_pop_results"] = int(rng.choice([1, 2, 100]))
        conf["planb"] = bool(rng.random() < 0.8)
        conf["epsilon"] = float(rng.choice([1e-3, 1e-1, 1e-7, 1e-12]))
        out = int(rng.integers(3))
        conf["output_MUUG"] = out != 1
        conf["output_haplotypes"] = out != 0
        conf["factor_missing_data"] = float(rng.choice([0.01, 0.0001]))
        if rng.random() < (0.5 if os.environ.get("GRIM_FUZZ_SAVE") else 0.1):
            conf["save_space_mode"] = True  # open_option_ cuts either operand to its ten largest entries (impute.py:1048-1059)
        pr = {"alpha": 0.4999999, "eta": 0, "beta": 1e-7, "gamma": 1e-7, "delta": 0.4999999}
        if rng.random() < 0.3:
            pr = {"alpha": float(rng.random()), "eta": float(rng.random() * 0.1), "beta": float(rng.random() * 0.2),
                  "gamma": float(rng.random() * 0.2), "delta": float(rng.random())}
        conf["priority"] = pr
        em = bool(rng.random() < 0.15) and conf["output_haplotypes"]
        gen = synth.SubjectGen(rows, int(rng.integers(1 << 30)), pops=pops)
        n = int(rng.integers(40, 160))
        lines = gen.mixed(n, amb=float(rng.random() * 0.7), miss=float(rng.random() * 0.4), recomb=float(rng.random() * 0.6))
        lines = [mutate(l, rng, gen.by_locus) if rng.random() < 0.35 else l for l in lines]
        binf = None
        if rng.random() < 0.2:  # per-subject phase masks (bin_imputation_in_file, impute.py:2001-2020)
            os.makedirs(harness.WORK, exist_ok=True)
            binf = os.path.join(harness.WORK, "fuzz_bin.json")
            ids = [l.split(",")[0] for l in lines]
            json.dump({sid: [int(x) for x in rng.integers(0, 2, 4)] for sid in ids[: max(1, len(ids) - 2)]}, open(binf, "w"))
            conf["bin_imputation_in_file"] = "data/subjects/fuzz_bin.json"
        if binf:
            conf["_bin_src"] = binf
        got, glog, imp = harness.run_product(gname, conf, lines, tag="fz", em_mr=em, on_unsupported="skip")
        if binf:
            conf["_bin_src"] = binf
        exp, elog = harness.run_oracle(gname, conf, lines, tag="fz_orc", em_mr=em)
        skipped = [sid for _, sid, _ in imp.unsupported]
        exp2 = harness.drop_subjects(exp, skipped)
        bad = [k for k in exp2 if exp2[k] != got[k]]
        print("round %3d %-4s thr=%-6d top=%-3d planb=%d out=%d em=%d bin=%d save=%d n=%-3d unsupported=%d %s  [%.0fs]" % (
            rd, gname, conf["number_of_options_threshold"], conf["max_haplotypes_number_in_phase"], conf["planb"], out, em, bool(binf),
            bool(conf.get("save_space_mode")), n,
            len(skipped), "OK" if not bad else "DIFF " + str(bad), time.time() - t0), flush=True)
        if bad:
            os.makedirs(os.path.join(harness.ROOT, "gpurun_out"), exist_ok=True)
            json.dump({"graph": gname, "conf": conf, "lines": lines, "em": em, "bad": bad},
                      open(os.path.join(harness.ROOT, "gpurun_out", "fuzz_fail.json"), "w"), indent=1)
            for k in bad:
                e = exp2[k].splitlines(); g = got[k].splitlines()
                for i in range(max(len(e), len(g))):
                    a = e[i] if i < len(e) else None; b = g[i] if i < len(g) else None
                    if a != b:
                        print("   ", k, "line", i, "\n      exp", a, "\n      got", b); break
            return 1
    print("fuzz: %d rounds identical" % rounds)
    return 0


if __name__ == "__main__":
    sys.exit(main())
