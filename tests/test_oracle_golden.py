"""The CPU oracle against every golden vector produced by the real reference (tools/make_golden.py)."""
import pytest

import harness


@pytest.mark.parametrize("scenario", harness.scenarios())
def test_oracle_matches_reference_outputs(scenario):
    gname, conf, lines, exp, elog, em = harness.golden(scenario)
    got, glog = harness.run_oracle(gname, conf, lines, tag="t_orc_" + scenario, em_mr=em)
    for k in exp:
        assert got[k] == exp[k], "%s: %s differs from the reference output" % (scenario, k)
    assert glog == elog, "%s: per-subject haplotype counts differ" % scenario


def test_known_answer_readme():
    """README.md:123-124 of the reference: D1 -> 8400 phased pairs, 6028 MUUGs."""
    gname, conf, lines, exp, elog, em = harness.golden("cau_min")
    assert elog == ["0 Subject: D1 8400 haplotypes", "0 Subject: D1 6028 haplotypes"]
    assert exp["umug"].splitlines()[0] == (
        "D1,A*01:02+A*02:01^B*15:01+B*15:01^C*03:03+C*03:04^DQB1*03:02+DQB1*06:02^DRB1*04:01+DRB1*15:01,"
        "8.838563003520004e-17,0")
