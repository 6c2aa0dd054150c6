#!/usr/bin/env python3
"""Run golden scenarios through the HIP path and diff against the reference outputs (debug aid)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import harness

def main():
    import __graft_entry__ as ge
    ge.build()
    only = [a for a in sys.argv[1:] if not a.startswith("-")]
    strict = "--strict" in sys.argv
    bad_total = 0
    for sc in harness.scenarios():
        if only and sc not in only:
            continue
        gname, conf, lines, exp, elog, em = harness.golden(sc)
        t = time.time()
        try:
            got, glog, imp = harness.run_product(gname, conf, lines, tag="g_" + sc, em_mr=em,
                                                 on_unsupported="raise" if strict else "skip")
        except Exception as e:
            print("%-18s ERROR %r" % (sc, e)); bad_total += 1; continue
        dt = time.time() - t
        skipped = [sid for _, sid, _ in imp.unsupported]
        exp2 = harness.drop_subjects(exp, skipped)
        bad = [k for k in exp2 if exp2[k] != got[k]]
        st = imp.last_stats
        print("%-18s %6.2fs  dev=%d deferred=%d kernel=%.2fms  %s" % (sc, dt, st.get("n", 0), len(skipped), st.get("kernel_ms", 0), "OK" if not bad else "MISMATCH " + str(bad)))
        for k in bad:
            e = exp2[k].splitlines(); g = got[k].splitlines()
            for i in range(max(len(e), len(g))):
                a = e[i] if i < len(e) else None; b = g[i] if i < len(g) else None
                if a != b:
                    print("   ", k, "line", i, "\n      exp", a, "\n      got", b); break
        bad_total += len(bad)
    print("TOTAL mismatching files:", bad_total)
    return 1 if bad_total else 0

if __name__ == "__main__":
    sys.path.insert(0, harness.ROOT)
    sys.exit(main())
