import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import harness, synth
sys.path.insert(0, harness.ROOT)
import numpy as np
os.environ["GRIM_QUIET"]="1"
os.environ.setdefault("GRIM_TIMING", "1")
rows = synth.read_freqs(synth.CAU_FREQS)
for pops, gname in ((["CAU"], "cau"), (harness.POPS["pop4"], "pop4")):
    gen = synth.SubjectGen(rows, 5, pops=pops)
    lines = gen.mixed(10000)
    conf = harness.base_conf(pops)
    from grim.imputation.impute import Imputation
    import grim.imputation.impute as I
    keep = {}
    orig = I.Imputation._run_arrays
    def spy(self, subj, tokens, priors, params):
        res, rws = orig(self, subj, tokens, priors, params); keep["res"]=res; keep["subj"]=subj; return res, rws
    I.Imputation._run_arrays = spy
    got, log, imp = harness.run_product(gname, conf, lines, tag="hist")
    I.Imputation._run_arrays = orig
    res = keep["res"]; subj = keep["subj"]
    nU = res["n_pairs"]; plan = res["plan"]
    print(gname, "kernel ms", imp.last_stats["kernel_ms"], imp.last_stats["kernel_a_ms"], imp.last_stats["kernel_b_ms"])
    for name, m in (("planA", plan==ord('a')), ("planB", plan==ord('b'))):
        v = nU[m]
        print(" ", name, "n=", m.sum(), "nU pct 50/90/99/max:", np.percentile(v,[50,90,99]).tolist(), v.max(), " <=64:", (v<=64).mean(), "<=256:", (v<=256).mean(), "<=1024:", (v<=1024).mean())
    nl = subj["n_loci"]; cand = subj["cnt"].astype(np.int64); 
    prod = np.ones(len(subj), dtype=np.int64)
    for l in range(5):
        mx = np.maximum(cand[:,l,0], cand[:,l,1]); mx[mx==0]=1; prod*=mx
    print("  n_loci hist", np.bincount(nl), " max-cand-per-side pct 50/90/99:", np.percentile(prod,[50,90,99]).tolist())
