"""Pins the oracle's caller with `-A` (keep all alternate alleles) on the reference's `call -mA -C alleles -T tab [-i]`
goldens (test.pl:289-297).  The constrained-alleles record rewriting is host logic restated in tests/helpers/cals.py;
what is checked of the engine is mcall() on the rewritten records: alleles kept, QUAL, AC/AN, GT, trimmed PL, DP4, MQ."""
import os
import pytest

from tests.helpers import orc, vcf, cals

CASES = [
    ("mpileup", "mpileup.cAls.out", "mpileup.tab", False), ("mpileup.2", "mpileup.cAls.2.out", "mpileup.2.tab", False),
    ("mpileup.3", "mpileup.cAls.3.out", "mpileup.3.tab", True), ("mpileup.3", "mpileup.cAls.4.out", "mpileup.4.tab", True),
    ("mpileup.3", "mpileup.cAls.5.out", "mpileup.5.tab", True), ("mpileup.4", "mpileup.cAls.6.out", "mpileup.6.tab", True),
    ("mpileup.5", "mpileup.cAls.7.out", "mpileup.7.tab", True),
    ("mpileup.cals.1", "mpileup.cals.8.out", "mpileup.cals.1.tab", False),      # an indel target paired with the SNP record
    ("mpileup.cals.2", "mpileup.cals.9.out", "mpileup.cals.2.tab", False),      # SNP and indel records at one position
]


def run_cals_case(G, idx, engine):
    inp, outp, tab, ins = CASES[idx]
    v = vcf.Vcf(os.path.join(G, inp + ".vcf"))
    g = vcf.Vcf(os.path.join(G, outp))
    out, names = cals.run(v, cals.parse_tab(os.path.join(G, tab)), engine, insert_missed=ins)
    assert cals.compare_with_golden(out, names, g) and len(g.recs) > 0


@pytest.mark.parametrize("idx", range(len(CASES)))
def test_oracle_reproduces_constrained_alleles_golden(golden_dir, idx):
    run_cals_case(os.path.join(golden_dir, "call"), idx, orc.mcall)
