#!/usr/bin/env python3
"""
tests/golden/intermediates/*.json: what the REAL reference computes INSIDE a subject, dumped through
class-level recording wrappers around its own methods (build container only; only data is committed):

  gen_phases                 the phases [H1, H2] of the subject, in order                               (impute.py:274-303)
  open_phases                per phase and side the candidate allele lists IN ORDER                      (impute.py:914-989)
  convert_list_to_one_dim    every top list [[p, [k, j]], ...] in call order                             (impute.py:424-442)
  call_comp_phase_prob       which plan answered (a / b / c) and the sizes of the two results

Candidate lists of more than a dozen entries and every top list are stored as their length, a SHA-256 of their repr and
their first and last entries; phases in full.  tests/test_oracle_golden.py records the same quantities from the oracle (same wrappers on the
oracle's methods) and compares, so a change that moves an intermediate fails at the intermediate, not at the output file.

    python tools/make_golden_intermediates.py
"""
import contextlib
import hashlib
import io
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
os.environ.setdefault("PYTHONHASHSEED", "0")
import make_golden as mg  # noqa: E402
import synth  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "intermediates")
FULL = 12  # candidate lists up to this many entries are stored whole


def digest(x, full=FULL):
    """a list -> itself when short, else {n, sha256 of repr, head, tail}"""
    if len(x) <= full:
        return x
    return {"n": len(x), "sha256": hashlib.sha256(repr(x).encode()).hexdigest(), "head": x[:2], "tail": x[-1:]}


def subjects():
    cau = synth.read_freqs(os.path.join(mg.GOLD, "data", "freqs", "CAU.freqs.gz"))
    a = synth.SubjectGen(cau, 71).mixed(24, amb=0.4, miss=0.25, recomb=0.3) + synth.SubjectGen(cau, 72).full(6) + synth.plan_c_cases("CAU")[:4]
    b = synth.SubjectGen(cau, 73, pops=mg.POP4).mixed(24, amb=0.35, miss=0.3, recomb=0.4)
    return a, b


def record(work, pops, lines, overrides, name):
    from grim.imputation.impute import Imputation
    from grim import grim

    rec = {"sid": None}
    per = {}

    def cur():
        return per.setdefault(rec["sid"], {"phases": None, "open": [], "top": [], "plan": None})

    orig = {k: getattr(Imputation, k) for k in ("impute_one", "gen_phases", "open_phases", "convert_list_to_one_dim")}

    def impute_one(self, subject_id, *a, **kw):
        rec["sid"] = subject_id
        out = orig["impute_one"](self, subject_id, *a, **kw)
        cur()["plan"] = self.plan
        return out

    def gen_phases(self, gen, n_loci, b_phases):
        out = orig["gen_phases"](self, gen, n_loci, b_phases)
        if cur()["phases"] is None:
            cur()["phases"] = [[list(h1), list(h2)] for h1, h2 in out]
        return out

    def open_phases(self, haps, N_Loc, gl_string):
        out = orig["open_phases"](self, haps, N_Loc, gl_string)
        cur()["open"].append([[digest([list(c) for c in side[0]]) for side in ph[:2]] for ph in out])
        return out

    def convert(self, prob):
        out = orig["convert_list_to_one_dim"](self, prob)
        cur()["top"].append(digest([[p, [int(k), int(j)]] for p, (k, j) in out], 2))
        return out

    Imputation.impute_one, Imputation.gen_phases = impute_one, gen_phases
    Imputation.open_phases, Imputation.convert_list_to_one_dim = open_phases, convert
    conf = dict(mg.BASE_CONF, populations=list(pops))
    conf.update(overrides)
    with open(os.path.join(work, "data", "subjects", "input.csv"), "w") as fh:
        fh.write("\n".join(lines) + "\n")
    with open(os.path.join(work, "conf.json"), "w") as fh:
        json.dump(conf, fh, indent=1)
    cwd = os.getcwd()
    os.chdir(work)
    try:
        with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
            grim.impute("conf.json")
    finally:
        os.chdir(cwd)
        for k, v in orig.items():
            setattr(Imputation, k, v)
    for v in per.values():  # a subject's top lists: their number, one digest over all of them, the first four as they are
        v["top"] = {"calls": len(v["top"]), "sha256": hashlib.sha256(repr(v["top"]).encode()).hexdigest(), "first": v["top"][:4]}
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, name + ".json"), "w") as fh:
        json.dump({"graph": os.path.basename(work), "conf": conf, "lines": lines, "subjects": per}, fh)
    tops = sum(v["top"]["calls"] for v in per.values())
    print("%-12s %d subjects, %d top lists, plans %s" % (name, len(per), tops, sorted({v["plan"] for v in per.values()})))


def main():
    mg.prepare_reference()
    sys.argv = ["x"]
    w1 = mg.build_graph("cau", ["CAU"])
    w4 = mg.build_graph("pop4", mg.POP4)
    a, b = subjects()
    record(w1, ["CAU"], a, {}, "cau")
    record(w4, mg.POP4, b, {"UNK_priors": "MR"}, "pop4")


if __name__ == "__main__":
    main()
