"""
Imputation driver for the MI355X engine.

Mirrors the reference's operator interface for this path (grim/imputation/impute.py):
`Imputation(net, config, count_by_prob)`, `impute_file(config, planb, em_mr, em)`,
`impute_one(...)`, the module functions `clean_up_gl`, `write_best_prob*` -- same argument meaning,
same six output files, same `.miss` / `.problem` rules -- but the per-subject work
(impute.py:1584-1724: candidate enumeration, graph look-ups, top-100, pair scoring, epsilon
ladder, ranking) runs as HIP kernels over the whole batch.  The host side only tokenises GL
strings to integers, builds the per-race prior matrices, and formats result rows.

There is no CPU fallback: without libgrim_hip.so and a HIP device `impute_file` raises.
"""

import logging
import os
import timeit

import numpy as np

from .. import _native as nat

# outcome of tokenising one input line
_DEV, _PROBLEM_ID, _PROBLEM_RAW, _MISS_NO_DEVICE, _UNSUPPORTED_GL = 0, 1, 2, 3, 5

REASONS = {
    2: "Plan B / Plan C fallback not resolved on the device (internal)",
    3: "Plan C for a subject whose '/'-lists stay above number_of_options_threshold after the most-common-allele reduction",
    5: "more than 4096 alleles in one GL string",
    8: "a GL string that names a locus twice or mixes loci in one entry (the reference pairs the entries by index, impute.py:246-272)",
}


def clean_up_gl(gl):
    """impute.py:105-118: every 'g' and 'L' character is deleted, loci whose text starts or ends
    with 'U' (the UUUU placeholders) are dropped."""
    gl = gl.replace("g", "").replace("L", "")
    return "^".join(part for part in gl.split("^") if part.strip("U") == part)


# ---- the reference's module-level writers (impute.py:24-99), for callers that format results themselves (the sibling
# EM package does): same arguments, same rows.  impute_file does not go through them -- the library's formatter
# threads produce the same text from the device's records.
def write_best_prob(name_gl, res, probs, numOfResult, fout, sign=","):
    """impute.py:24-58: pairs (a, b) and (b, a) are one entry, printed in the orientation seen first; probabilities
    added in list order; stable sort, bigger first; the best numOfResult rows."""
    total = {}
    for (a, b), p in zip(res, probs):
        key = a + sign + b
        if key not in total:
            swapped = b + sign + a
            if swapped in total:
                key = swapped
        total[key] = p + total[key] if key in total else p
    ranked = sorted(total.items(), key=lambda kv: kv[1], reverse=True)
    for k, (key, p) in enumerate(ranked[:numOfResult]):
        fout.write(name_gl + "," + str(key) + "," + str(p) + "," + str(k) + "\n")


def write_best_prob_genotype(name_gl, res, numOfResult, fout):
    """impute.py:61-76: {genotype: probability} -> the best numOfResult rows, stable on ties."""
    ranked = sorted(res.items(), key=lambda kv: kv[1], reverse=True)
    for k, (gl, p) in enumerate(ranked[:numOfResult]):
        fout.write(name_gl + "," + str(gl) + "," + str(p) + "," + str(k) + "\n")


def write_best_hap_race_pairs(name_gl, haps, pops, probs, numOfResult, fout):
    """impute.py:79-99: every (haplotype pair, race pair) its own row "h1;r1,h2;r2", stable sort, bigger first."""
    rows = [(probs[i], haps[i][0] + ";" + pops[i][0] + "," + haps[i][1] + ";" + pops[i][1]) for i in range(len(probs))]
    rows.sort(key=lambda r: r[0], reverse=True)
    for k, (p, pair) in enumerate(rows[:numOfResult]):
        fout.write(name_gl + "," + str(pair) + "," + str(p) + "," + str(k) + "\n")


class UnsupportedSubjects(NotImplementedError):
    def __init__(self, items):
        self.items = items
        kinds = sorted({r for _, _, r in items})
        msg = "%d subject(s) need a reference path this build does not run on the GPU yet: %s (first ids: %s)" % (
            len(items), "; ".join(REASONS.get(k, str(k)) for k in kinds), ", ".join(str(s) for _, s, _ in items[:5]))
        super().__init__(msg)


class Imputation(object):
    def __init__(self, net=None, config=None, count_by_prob=None, verbose=False, device=None):
        self.logger = logging.getLogger("Logger." + __name__)
        self.verbose = verbose
        self.netGraph = net
        self.device = device
        self.quiet = bool(int(os.environ.get("GRIM_QUIET", "0")))
        self.on_unsupported = os.environ.get("GRIM_ON_UNSUPPORTED", "raise")
        self.unsupported = []
        self.last_stats = {}
        if config is None:
            return
        self.config = config
        self.populations = config["pops"]
        P = len(self.populations)
        if P > nat.MAXPOP:
            raise NotImplementedError("more than %d populations" % nat.MAXPOP)
        self.unk_priors = config["UNK_priors"]
        self.full_loci = config["full_loci"]
        self.index_dict = dict(config["loci_map"])
        if config.get("nodes_for_plan_A"):
            raise NotImplementedError("Plan_A_Matrix is not supported by this build")
        if count_by_prob is None:  # impute.py:205-212
            self.count_by_prob = np.ones(P)
            if config["use_pops_count_file"]:
                with open(config["pops_count_file"]) as fh:
                    for i, line in enumerate(fh):
                        self.count_by_prob[i] = float(line.strip().split(",")[2])
        else:
            self.count_by_prob = count_by_prob
        self.plan = "a"
        self.option_1 = 0
        self.option_2 = 0
        self._prior_cache = {}
        self._priors = []

    # ---- prior matrix: impute.py:1844-1924, evaluated once per distinct race pair -----------------
    def _prior_matrix(self, race1, race2, priority):
        P = len(self.populations)
        base = np.ones((P, P)) if self.unk_priors == "MR" else np.identity(P)
        if not (race1 or race2):
            return base
        list1 = race1.split(";")
        list2 = race2.split(";")
        known = False
        for lst in (list1, list2):
            for i, r in enumerate(lst):
                if r in self.populations:
                    known = True
                else:
                    lst[i] = ""
        if not known:
            return base
        acc = np.zeros((P, P))
        eye = np.identity(P)
        g, al, de = priority["gamma"], priority["alpha"], priority["delta"]
        for ra in list1:
            for rb in list2:
                if ra == "" and rb == "":
                    continue
                t = np.zeros((P, P))
                if ra == "" or rb == "":
                    r = self.populations.index(rb if ra == "" else ra)
                    t[r, :] = t[r, :] + g * 2
                    t = t + t.T
                    t[r, r] -= g * 2
                else:
                    a, b = self.populations.index(ra), self.populations.index(rb)
                    for i in range(P):  # row and column may overlap at (a,b): keep the scalar order
                        t[a, i] = t[a, i] + g
                        t[i, b] = t[i, b] + g
                    t[a, b] -= g
                    t[a, b] = t[a, b] + al
                    if a != b:
                        t = t + t.T
                        t[a, a] -= g
                        t[b, b] -= g
                    t[a, a] += de
                    if a != b:
                        t[b, b] += de
                t = priority["eta"] * np.ones((P, P)) + t + priority["beta"] * eye
                acc += t
        total = 0
        for i in range(P):
            for j in range(P):
                acc[i][j] = acc[i][j] * self.count_by_prob[i] * self.count_by_prob[j]
                total += acc[i][j]
        return acc / total

    def _prior_index(self, race1, race2, priority):
        key = (race1, race2)
        idx = self._prior_cache.get(key)
        if idx is None:
            m = self._prior_matrix(race1, race2, priority)
            idx = len(self._priors)
            if idx >= 65535:
                raise OverflowError("more than 65535 distinct race pairs in one batch")
            self._priors.append(np.ascontiguousarray(m, dtype=np.float64))
            self._prior_cache[key] = idx
        return idx

    # ---- GL string -> integer tokens (impute.py:105-118, 246-272) ----------------------------------
    def _tokenise(self, gl, planb):
        """-> (kind, payload).  payload for _DEV: (n_loci, slots, same_mask, [(ids0, w0, ids1, w1)...])."""
        g = self.netGraph
        cleaned = clean_up_gl(gl)
        if not gl:
            return _PROBLEM_ID, None
        if cleaned == "" or cleaned == " ":
            return _PROBLEM_ID, None
        side1, side2, blanks = [], [], 0
        parts = cleaned.split("^")
        for p in parts:
            if p[0] == "+":  # IndexError for an empty entry -> caller's except -> raw line
                p = p[1:]
            two = p.split("+")
            if len(two) == 1:
                if two == [""]:
                    blanks += 1
                    continue
                return _PROBLEM_ID, None
            side1.append(two[0])
            side2.append(two[1])
        n = len(parts) - blanks
        side1.sort()
        side2.sort()
        if n != len(side1) or n < 1:
            raise ValueError("irregular GL string")
        if n > len(self.full_loci):
            return _UNSUPPORTED_GL, None  # more entries than the graph has loci: the reference opens 2^(n-1) phases all the same
        slots, same, pos, unknown_locus = [], 0, [], False
        for k in range(n):
            alts1, alts2 = side1[k].split("/"), side2[k].split("/")
            loci = {a.split("*")[0] for a in alts1} | {a.split("*")[0] for a in alts2}
            if len(loci) != 1:
                return _UNSUPPORTED_GL, None  # reason 8: the reference goes on with what gl2haps paired by index
            locus = loci.pop()
            if locus not in g.locus_slot:
                unknown_locus = True
                continue
            s = g.locus_slot[locus]
            if s in slots:
                return _UNSUPPORTED_GL, None
            slots.append(s)
            if side1[k] == side2[k]:
                same |= 1 << k
            ent = []
            for alts in (alts1, alts2):
                seen, ids = set(), []
                for a in alts:
                    if a not in seen:
                        seen.add(a)
                        ids.append(g.allele_id(s, a))
                ent.append((ids, min(len(alts), 65535)))
            pos.append(ent)
        if unknown_locus:
            # a locus name outside loci_map: Plan A cannot match, Plan B dies with KeyError in
            # check_if_alleles_exist (impute.py:1218-1222) -> bare except -> raw line in .problem;
            # without Plan B the subject is a plain miss.
            if planb:
                raise KeyError("locus not in loci_map")
            return _MISS_NO_DEVICE, None
        return _DEV, (n, slots, same, pos)

    # ---- parameters ---------------------------------------------------------------------------------
    def _params(self, config, planb, em_mr, em=False):
        p = nat.Params()
        eps = config["epsilon"]
        ladder = []
        while eps > 0:  # impute.py:1665-1673
            eps /= 10
            if eps < 1.0e-9:
                eps = 0.0
            ladder.append(eps)
            if len(ladder) > nat.MAXLADDER:
                raise NotImplementedError("epsilon ladder longer than %d steps" % nat.MAXLADDER)
        for i, e in enumerate(ladder):
            p.ladder[i] = e
        p.n_ladder = len(ladder)
        p.top_n = int(config["max_haplotypes_number_in_phase"])
        if p.top_n > nat.TOPCAP:
            raise NotImplementedError("max_haplotypes_number_in_phase > %d" % nat.TOPCAP)
        p.opt_threshold = int(config["number_of_options_threshold"])
        p.n_results = int(config["number_of_results"])
        p.n_pop_results = int(config["number_of_pop_results"])
        p.out_muug = 1 if config["output_MUUG"] else 0
        p.out_haps = 1 if config["output_haplotypes"] else 0
        p.planb = 1 if planb else 0
        p.em = 1 if em else 0
        p.em_mr = 1 if em_mr else 0
        p.eps_nonpositive = 0 if config["epsilon"] > 0 else 1
        p.save_mode = 1 if config.get("save_mode") else 0
        order = sorted(range(len(self.populations)), key=lambda i: self.populations[i])
        for rank, i in enumerate(order):
            p.pop_rank[i] = rank
        for k in range(nat.MAXL + 1):
            p.factor_missing_pow[k] = config["factor_missing_data"] ** k
        rows = config["matrix_planb"]
        if len(rows) > nat.MAXROWS:
            raise NotImplementedError("Plan_B_Matrix with more than %d rows" % nat.MAXROWS)
        p.planb_rows = len(rows)
        for r, row in enumerate(rows):
            if len(row) > nat.MAXL:
                raise NotImplementedError("Plan_B_Matrix row with more than %d blocks" % nat.MAXL)
            p.planb_nblk[r] = len(row)
            for b, blk in enumerate(row):
                m = 0
                for idx in blk:
                    m |= 1 << self.full_loci.index(str(idx))
                p.planb_blk[r][b] = m
        return p

    # ---- batch on the device --------------------------------------------------------------------------
    def run_batch(self, records, config, planb, em_mr=False, keep=False, em=False):
        """records: list of (n_loci, slots, same_mask, positions, prior_idx).  Returns (res, rows)."""
        n = len(records)
        subj = np.zeros(n, dtype=nat.SUBJECT_DT)
        toks = []
        off = 0
        for i, (nl, slots, same, pos, pidx) in enumerate(records):
            s = subj[i]
            s["tok_off"] = off
            s["prior_idx"] = pidx
            s["n_loci"] = nl
            s["pad"][0] = same
            for k in range(nl):
                s["slot"][k] = slots[k]
                for side in range(2):
                    ids, wid = pos[k][side]
                    s["cnt"][k][side] = len(ids)
                    s["wid"][k][side] = wid
                    toks.extend(ids)
                    off += len(ids)
        tokens = np.array(toks if toks else [0], dtype=np.uint16)
        priors = np.stack(self._priors) if self._priors else np.ones((1, len(self.populations), len(self.populations)))
        ctx = nat.default_context(self.device)
        dgraph = self.netGraph.device(ctx)
        params = self._params(config, planb, em_mr, em)
        tu = timeit.default_timer()
        batch = nat.DeviceBatch(ctx, dgraph, params, subj, tokens, priors)
        t0 = timeit.default_timer()
        batch.run()
        t1 = timeit.default_timer()
        res, rows = batch.results()
        t2 = timeit.default_timer()
        self.last_stats = {
            "upload_s": t0 - tu, "download_s": t2 - t1,
            "n": n, "run_s": t1 - t0, "kernel_ms": batch.kernel_ms(0), "kernel_a_ms": batch.kernel_ms(1),
            "kernel_b_ms": batch.kernel_ms(2), "counters": batch.counters(),
        }
        if keep:
            return res, rows, batch
        batch.close()
        return res, rows

    # ---- formatting ------------------------------------------------------------------------------------
    def _genotype(self, key_a, key_b):
        """MUUG text of a haplotype pair (impute.py:497-504)."""
        g = self.netGraph
        a = sorted(x for x in g.key_alleles(key_a) if x)
        b = sorted(x for x in g.key_alleles(key_b) if x)
        return "^".join("+".join(sorted(z)) for z in zip(a, b))

    def _hap_name(self, key, plan=None):
        """a haplotype as the reference's phased writer spells it: a graph node's name (Plan A; a Plan-B list answered by one
        look-up: bit 60 of the key) in the graph's locus order, a joined key with its alleles sorted (impute.py:1041-1069)"""
        alleles = [x for x in self.netGraph.key_alleles(key) if x]
        if plan == ord("a") or (int(key) >> 60) & 1:
            return "~".join(alleles)
        return "~".join(sorted(alleles))

    def _pop_name(self, idx, plan):
        return "all_pops" if plan == ord("c") else self.populations[int(idx)]

    # ---- reference-shaped per-subject API (impute.py:1940-1983) -----------------------------------------
    def impute_one(self, subject_id, gl, binary, race1, race2, priority, epsilon, n, MUUG_output, haps_output,
                   planb, em):
        """Runs ONE subject through the device (a batch of one).  Returns (id, res_muugs, res_haps) in
        the reference's dict shapes.  `number_of_results` style truncation is NOT applied here; the
        dicts carry at most config['number_of_results'] best entries per table (the device ranks)."""
        cfg = dict(self.config)
        cfg["epsilon"] = epsilon
        cfg["output_MUUG"] = MUUG_output
        cfg["output_haplotypes"] = haps_output
        self._prior_cache, self._priors = {}, []
        kind, payload = self._tokenise(gl, planb)
        if kind == _PROBLEM_ID:
            return subject_id, None, None
        res_m = {"MaxProb": 0, "Haps": {}, "Pops": {}}
        res_h = {"Haps": [], "Probs": [], "Pops": []}
        if kind == _UNSUPPORTED_GL:
            raise UnsupportedSubjects([(0, subject_id, 8)])
        if kind == _MISS_NO_DEVICE:
            return subject_id, res_m, res_h
        pidx = self._prior_index(race1 or "", race2 or "", priority)
        res, rows = self.run_batch([payload + (pidx,)], cfg, planb, em=em)
        r = res[0]
        if r["status"] == nat.ST_UNSUPPORTED:
            raise UnsupportedSubjects([(0, subject_id, int(r["reason"]))])
        plan = int(r["plan"])
        self.plan = chr(plan) if plan else "a"
        if MUUG_output:
            res_m["MaxProb"] = float(r["max_prob"])
            for row in rows[r["row_off"][nat.T_UMUG]: r["row_off"][nat.T_UMUG] + r["n_rows"][nat.T_UMUG]]:
                res_m["Haps"][self._genotype(row["a"], row["b"])] = float(row["prob"])
            for row in rows[r["row_off"][nat.T_UMUG_POPS]: r["row_off"][nat.T_UMUG_POPS] + r["n_rows"][nat.T_UMUG_POPS]]:
                res_m["Pops"][self._pop_name(row["popa"], plan) + "," + self._pop_name(row["popb"], plan)] = float(row["prob"])
        if haps_output:
            plan = int(r["plan_phased"]) or plan
            res_h["MaxProb"] = float(r["max_prob"])
            for row in rows[r["row_off"][nat.T_PMUG]: r["row_off"][nat.T_PMUG] + r["n_rows"][nat.T_PMUG]]:
                res_h["Haps"].append([self._hap_name(row["a"], plan), self._hap_name(row["b"], plan)])
                res_h["Probs"].append(float(row["prob"]))
                res_h["Pops"].append([self._pop_name(row["popa"], plan), self._pop_name(row["popb"], plan)])
        return subject_id, res_m, res_h

    # ---- file driver (impute.py:1985-2155) ---------------------------------------------------------------
    # ---- EM hook of the sibling EM package (impute.py:305-351): pure string work on the host ------------
    @staticmethod
    def gl2haps(GL_String):
        """impute.py:246-272: {"Genotype": [sorted side-1 entries, sorted side-2 entries], "N_Loc": n}, or []."""
        if GL_String in ("", " "):
            return []
        first, second, n_loci = [], [], 0
        for locus in GL_String.split("^"):
            if locus[0] == "+":  # an empty entry raises IndexError here, as in the reference
                locus = locus[1:]
            halves = locus.split("+")
            if len(halves) == 1:
                if halves[0] == "":
                    continue
                return []
            first.append(halves[0])
            second.append(halves[1])
            n_loci += 1
        return {"Genotype": [sorted(first), sorted(second)], "N_Loc": n_loci}

    @staticmethod
    def gen_phases(gen, n_loci, b_phases=None):
        """impute.py:274-303: the <= 2^(n-1) phases [H1, H2], mirror images and repeats dropped."""
        free = None if b_phases is None else {k for k, bit in enumerate(b_phases) if bit == 1}
        phases, seen = [], set()
        for code in range(1 << (n_loci - 1)):
            side = [(code >> k) & 1 if (free is None or k in free) else 0 for k in range(n_loci)]
            h1 = [gen[side[k]][k] for k in range(n_loci)]
            h2 = [gen[1 - side[k]][k] for k in range(n_loci)]
            a, b = "~".join(h1), "~".join(h2)
            if (a + "^" + b) not in seen or (b + "^" + a) not in seen:
                seen.update((a + "^" + b, b + "^" + a))
                phases.append([h1, h2])
        return phases

    @staticmethod
    def open_phases_for_em(haps, N_Loc, cutoff):
        """impute.py:322-351: per phase [[haplotypes of side 1], [haplotypes of side 2]] with every '/'
        ambiguity expanded (position 0 most significant); a phase with a side of >= cutoff options is dropped."""
        import itertools

        out = []
        for phase in haps:
            sides = []
            for entries in phase[:2]:
                alts = [tuple(e.split("/")) for e in entries]
                options = 1
                for i in range(N_Loc):
                    options *= len(alts[i])
                if options < cutoff:
                    sides.append([[list(c) for c in itertools.product(*alts[:N_Loc])]])
                else:
                    sides.append([])
            if sides[0] and sides[1]:
                out.append(sides)
        return out

    def open_gl_string(self, gl_string, cutoff):
        """impute.py:305-320."""
        chrom = self.gl2haps(gl_string)
        if chrom == []:
            return None
        phases = self.gen_phases(chrom["Genotype"], chrom["N_Loc"], None)
        if phases == []:
            return None
        return self.open_phases_for_em(phases, chrom["N_Loc"], cutoff)

    def impute_file(self, config, planb=None, em_mr=False, em=False):
        """impute.py:1985-2155.  The input file is read, tokenised, imputed, formatted and written by the library's
        streaming pipeline (grim_stream_*: chunks of lines, tokenizer threads -> device -> formatter threads ->
        ordered pwrite), so no per-line Python work remains; Python prints the per-subject lines when not quiet."""
        out_paths = {}
        for key, path_key, flag in self._OUT_FILES:
            if flag is None or config[flag]:
                out_paths[key] = config[path_key]
        self._stream_run(config, planb, em_mr, em, out_paths=out_paths, in_path=config["imputation_input_file"])

    _OUT_FILES = [("umug", "imputation_out_umug_freq_file", "output_MUUG"),
                  ("umug_pops", "imputation_out_umug_pops_file", "output_MUUG"),
                  ("pmug", "imputation_out_hap_freq_file", "output_haplotypes"),
                  ("pmug_pops", "imputation_out_hap_pops_file", "output_haplotypes"),
                  ("miss", "imputation_out_miss_file", None), ("problem", "imputation_out_problem_file", None)]

    @classmethod
    def write_outputs(cls, config, texts):
        for key, path_key, flag in cls._OUT_FILES:
            if flag is not None and not config[flag]:
                continue
            data = texts[key]
            with open(config[path_key], "wb" if isinstance(data, bytes) else "w") as fh:
                fh.write(data)

    def impute_lines(self, lines, config, planb=None, em_mr=False, line_offset=0, em=False, as_bytes=False):
        """The body of impute_file on a list of input lines.  Returns the six output texts keyed
        'umug','umug_pops','pmug','pmug_pops','miss','problem'.  `line_offset` is the global index
        of lines[0] (multi-GPU shards keep the reference's line numbers in .miss/.problem)."""
        data = "".join(l if l.endswith("\n") else l + "\n" for l in lines).encode()
        return self._stream_run(config, planb, em_mr, em, data=data, line_offset=line_offset, as_bytes=as_bytes)

    @staticmethod
    def _phase_masks(config):
        """phase masks (impute.py:2001-2005, 2030-2032, 277-290): position m may switch sides only where the subject's list
        holds 1; an id missing from the file is a KeyError in the reference -> raw line to .problem.  None: no mask file."""
        if not os.path.isfile(config["bin_imputation_input_file"]):
            return None
        import json
        with open(config["bin_imputation_input_file"]) as fh:
            f_bin = json.load(fh)
        masks = {}
        for sid, mask in f_bin.items():
            fixed = 0
            for m in range(nat.MAXL):
                if not (m < len(mask) and mask[m] == 1):
                    fixed |= 1 << m
            masks[sid] = fixed
        return masks

    def _stream_run(self, config, planb, em_mr, em, out_paths=None, in_path=None, data=None, line_offset=0, as_bytes=False):
        if planb is None:
            planb = config["planb"]
        self.unsupported = []
        masks = self._phase_masks(config)
        params = self._params(config, planb, em_mr, em)
        ps, keep = nat.prior_spec(config["priority"], self.unk_priors, self.count_by_prob)
        ctx = nat.default_context(self.device)
        dgraph = self.netGraph.device(ctx)
        t0 = timeit.default_timer()
        st = nat.Stream(ctx, dgraph, self.netGraph.adict, params, ps, self.populations, out_paths=out_paths,
                        want_log=not self.quiet, line_offset=line_offset, masks=masks, timing=bool(int(os.environ.get("GRIM_TIMING", "0"))))
        try:
            if in_path is not None:
                st.write_file(in_path)
            elif data:
                # the lines are in memory and stay there until the stream is closed: lent, not copied (grim_stream_write_borrowed)
                st.write(data, borrowed=isinstance(data, bytes))
            st.finish()
            stats = st.stats()
            self.last_stats = {
                "n": int(stats.subjects), "lines": int(stats.lines), "chunks": int(stats.chunks), "reruns": int(stats.reruns),
                "wall_s": stats.wall_s, "total_s": timeit.default_timer() - t0, "device_s": stats.device_s,
                "kernel_ms": stats.kernel_ms[0], "kernel_a_ms": stats.kernel_ms[1], "kernel_b_ms": stats.kernel_ms[2],
                "counters": [int(x) for x in stats.counters],
                "host_s": {"tokenize_cpu": stats.tokenize_cpu_s, "format_cpu": stats.format_cpu_s, "write_cpu": stats.write_cpu_s},
                "text_bytes": [int(x) for x in stats.text_bytes],
            }
            self.unsupported = st.unsupported()
            if not self.quiet:
                import sys
                sys.stdout.write(st.text(6))
            texts = None
            if out_paths is None:
                texts = {key: st.text(k, as_bytes=as_bytes) for k, key in enumerate(nat.TEXT_KEYS)}
        finally:
            st.close()
        if self.unsupported and self.on_unsupported == "raise":
            raise UnsupportedSubjects(self.unsupported)
        return texts

    def impute_lines_block(self, lines, config, planb=None, em_mr=False, line_offset=0, em=False):
        """The same through the block entry points of the C-ABI (grim_tokenize -> grim_batch_upload/run/results ->
        grim_format: the whole input as ONE device batch, every step on the calling thread).  The streaming
        pipeline is the product path; this one is what a caller with its own scheduling would use, and the tests
        hold the two against each other."""
        priority = config["priority"]
        if planb is None:
            planb = config["planb"]
        self.unsupported = []
        text = "".join(l if l.endswith("\n") else l + "\n" for l in lines).encode()
        parsed = nat.Parsed(self.netGraph.adict, text, planb)
        try:
            races = parsed.races()
            P = len(self.populations)
            ps, keep = nat.prior_spec(priority, self.unk_priors, self.count_by_prob)
            priors = np.ones((max(1, len(races)), P, P))
            for k, (r1, r2) in enumerate(races):
                priors[k] = nat.prior_matrix(ps, self.populations, r1, r2)
            subj = parsed.subjects()
            kinds = parsed.kinds()
            dev = parsed.dev_index()
            params = self._params(config, planb, em_mr, em)
            if len(subj):
                res, rows = self._run_arrays(subj, parsed.tokens(), priors, params)
            else:
                res, rows = np.zeros(0, dtype=nat.RESULT_DT), np.zeros(0, dtype=nat.ROW_DT)
            host_reason = {nat.K_UNSUPPORTED: 5, nat.K_UNSUPPORTED_GL: 8}
            bad_lines = [int(j) for j in range(len(kinds)) if kinds[j] in host_reason or
                         (kinds[j] == nat.K_DEVICE and res["status"][dev[j]] == nat.ST_UNSUPPORTED)]
            self.unsupported = [(line_offset + j, parsed.subject_id(j), host_reason[kinds[j]] if kinds[j] in host_reason else int(res[dev[j]]["reason"]))
                                for j in bad_lines]
            if self.unsupported and self.on_unsupported == "raise":
                raise UnsupportedSubjects(self.unsupported)
            return parsed.format(self.netGraph.adict, params, self.populations, res, rows, line_offset, None)
        finally:
            parsed.close()

    def _run_arrays(self, subj, tokens, priors, params):
        ctx = nat.default_context(self.device)
        dgraph = self.netGraph.device(ctx)
        tu = timeit.default_timer()
        batch = nat.DeviceBatch(ctx, dgraph, params, subj, tokens, priors)
        t0 = timeit.default_timer()
        batch.run()
        t1 = timeit.default_timer()
        res, rows = batch.results()
        t2 = timeit.default_timer()
        self.last_stats = {
            "upload_s": t0 - tu, "download_s": t2 - t1,
            "n": len(subj), "run_s": t1 - t0, "kernel_ms": batch.kernel_ms(0), "kernel_a_ms": batch.kernel_ms(1),
            "kernel_b_ms": batch.kernel_ms(2), "counters": batch.counters(),
        }
        batch.close()
        return res, rows

    def impute_lines_python(self, lines, config, planb=None, em_mr=False, line_offset=0, em=False):
        """The same in pure Python host code (tokeniser `_tokenise`, formatter `_write_rows`): kept as
        the cross-check of the C++ host helpers in tests/ and as documentation of their rules."""
        priority = config["priority"]
        muug_on = config["output_MUUG"]
        haps_on = config["output_haplotypes"]
        if planb is None:
            planb = config["planb"]
        self._prior_cache, self._priors = {}, []
        self.unsupported = []

        outcome = []  # per line: (kind, subject_id, raw line, device index)
        records = []
        for raw in lines:
            line = raw.rstrip()
            sid = None
            try:
                parts = line.split(",") if "," in line else line.split("%")
                sid = parts[0]
                gl = parts[1]
                race1 = race2 = None
                if len(parts) > 2:
                    race1, race2 = parts[2], parts[3]
                pidx = self._prior_index(race1 or "", race2 or "", priority)
                kind, payload = self._tokenise(gl, planb)
                if kind == _DEV:
                    outcome.append((_DEV, sid, line, len(records)))
                    records.append(payload + (pidx,))
                else:
                    outcome.append((kind, sid, line, -1))
            except Exception:  # the reference's bare except (impute.py:2141-2144)
                outcome.append((_PROBLEM_RAW, sid, line, -1))

        if records:
            res, rows = self.run_batch(records, config, planb, em_mr, em=em)
        else:
            res, rows = np.zeros(0, dtype=nat.RESULT_DT), np.zeros(0, dtype=nat.ROW_DT)

        bad = [(line_offset + i, outcome[i][1], 8 if outcome[i][0] == _UNSUPPORTED_GL else int(res[outcome[i][3]]["reason"]))
               for i in range(len(outcome))
               if outcome[i][0] == _UNSUPPORTED_GL or (outcome[i][0] == _DEV and res[outcome[i][3]]["status"] == nat.ST_UNSUPPORTED)]
        self.unsupported = bad
        if bad and self.on_unsupported == "raise":
            raise UnsupportedSubjects(bad)
        skip = {i for i, _, _ in bad}

        out = {k: [] for k in ("umug", "umug_pops", "pmug", "pmug_pops", "miss", "problem")}
        for j, (kind, sid, line, di) in enumerate(outcome):
            i = line_offset + j
            if i in skip:
                continue
            if kind == _PROBLEM_RAW:
                out["problem"].append(str(line) + "\n")
                continue
            if kind == _PROBLEM_ID:
                out["problem"].append(str(i) + "," + str(sid) + "\n")
                continue
            if kind == _MISS_NO_DEVICE:
                n_pairs = n_geno = 0
                r = None
            else:
                r = res[di]
                if r["status"] == nat.ST_NOPHASE:  # impute.py:1607-1609 placeholder -> raises in the phased writer
                    if haps_on:
                        out["problem"].append(str(line) + "\n")
                    continue
                n_pairs = int(r["n_pairs"]) if haps_on else 0
                n_geno = int(r["n_genotypes"]) if muug_on else 0
            # impute.py:2065-2068 -- with output_haplotypes off res_haps["Haps"] is the 3-char
            # placeholder "Nan", whose len() is not 0, so .miss is never written then
            if haps_on and n_pairs == 0 and n_geno == 0:
                out["miss"].append(str(i) + "," + str(sid) + "\n")
            plan = int(r["plan"]) if r is not None else ord("a")
            if haps_on and r is not None:
                plan_h = int(r["plan_phased"]) or plan
                self._write_rows(out["pmug"], sid, rows, r, nat.T_PMUG, plan_h, em_mr)
                self._write_rows(out["pmug_pops"], sid, rows, r, nat.T_PMUG_POPS, plan_h, em_mr)
            if muug_on and r is not None:
                self._write_rows(out["umug"], sid, rows, r, nat.T_UMUG, plan, em_mr)
                self._write_rows(out["umug_pops"], sid, rows, r, nat.T_UMUG_POPS, plan, em_mr)
                if plan == ord("c") and int(r["n_rows"][nat.T_UMUG_POPS]) == 0:
                    # Plan C always reports {"all_pops,all_pops": sum(...)}, an integer 0 when it
                    # found nothing (impute.py:1375-1378)
                    out["umug_pops"].append(sid + ",all_pops,all_pops,0,0\n")
        return {k: "".join(v) for k, v in out.items()}

    def _write_rows(self, fh, sid, rows, r, table, plan, em_mr):
        a0 = int(r["row_off"][table])
        for k in range(int(r["n_rows"][table])):
            row = rows[a0 + k]
            prob = float(row["prob"])
            if table == nat.T_UMUG:
                text = self._genotype(row["a"], row["b"])
            elif table == nat.T_PMUG:
                if em_mr:  # write_best_hap_race_pairs, impute.py:79-99
                    text = (self._hap_name(row["a"], plan) + ";" + self._pop_name(row["popa"], plan) + "," +
                            self._hap_name(row["b"], plan) + ";" + self._pop_name(row["popb"], plan))
                else:
                    text = self._hap_name(row["a"], plan) + "+" + self._hap_name(row["b"], plan)
            else:
                text = self._pop_name(row["popa"], plan) + "," + self._pop_name(row["popb"], plan)
            fh.append(sid + "," + text + "," + str(prob) + "," + str(k) + "\n")
