"""Pins the oracle's multiallelic caller against every `call -m` golden of the reference's test-suite
(test.pl:276-288,303-308): pooled calling, -S ploidy lists, --ploidy-file with sexes/PED, -G groups
(per-sample and file), -F prior frequencies, GQ/GP."""
import os
import pytest

from bcftools_amd import abi
from tests.helpers import orc, vcf, calldrv as D

V = abi.CALL_VARONLY


def groups_file(p):
    d = {}
    for l in open(p):
        f = l.split()
        if len(f) >= 2:
            d[f[0]] = f[1]
    return d


def cases(G):
    j = lambda f: os.path.join(G, f)
    return [
        ("mpileup", "mpileup.1.out", dict(call_flag=V)),
        ("mpileup", "mpileup.3.out", dict(call_flag=V, samples=D.parse_samples_file(j("mpileup.3.samples")))),
        ("mpileup", "mpileup.4.out", dict(call_flag=V, samples=D.parse_samples_file(j("mpileup.4.samples")))),
        ("mpileup", "mpileup.5.out", dict(call_flag=V, samples=D.parse_samples_file(j("mpileup.5.samples")))),
        ("mpileup.X", "mpileup.X.out", dict(call_flag=V, samples=D.parse_samples_file(j("mpileup.samples")),
                                            ploidy=D.parse_ploidy_file(j("mpileup.ploidy")))),
        ("mpileup.X", "mpileup.X.out", dict(call_flag=V, samples=D.parse_samples_file(j("mpileup.ped")),
                                            ploidy=D.parse_ploidy_file(j("mpileup.ploidy")))),
        ("mpileup.X", "mpileup.X.2.out", dict(call_flag=V, samples=D.parse_samples_file(j("mpileup.2.samples")),
                                              ploidy=D.parse_ploidy_file(j("mpileup.ploidy")))),
        ("mpileup.NA19213.NA19129", "mpileup.hwe.1.out", dict(call_flag=V)),
        ("mpileup.NA19213.NA19129", "mpileup.hwe.1b.out", dict(call_flag=V, groups="-", grp_tag="AD")),
        ("mpileup.hwe", "mpileup.hwe.2.out", dict(call_flag=V)),
        ("mpileup.hwe", "mpileup.hwe.3.out", dict(call_flag=V, groups="-", grp_tag="AD")),
        ("mpileup.hwe", "mpileup.hwe.4.out", dict(call_flag=V, groups=groups_file(j("mpileup.hwe.samples")), grp_tag="AD")),
        ("call-G", "call-G.1.out", dict(call_flag=V)),
        ("call-G", "call-G.2.out", dict(call_flag=V, groups="-", grp_tag="AD")),
        ("call-G.2", "call-G.2.1.out", dict(call_flag=V, prior=("AN_POP", "AC_POP"))),
        ("call.af-fixation", "call.af-fixation.1.out", dict()),
        ("call.af-fixation", "call.af-fixation.2.out", dict(groups=groups_file(j("call.af-fixation.txt")))),
        ("call.af-fixation", "call.af-fixation.3.out", dict(groups=groups_file(j("call.af-fixation.txt")),
                                                            output_tags=abi.CALL_FMT_GQ | abi.CALL_FMT_GP)),
    ]


N_CASES = 18


def run_case(G, idx, engine):
    inp, outp, kw = cases(G)[idx]
    v = vcf.Vcf(os.path.join(G, inp + ".vcf"))
    g = vcf.Vcf(os.path.join(G, outp))
    called, names = D.run_call(v, engine, **kw)
    tags = ("GQ", "GP") if kw.get("output_tags") else ()
    D.compare_with_golden(called, names, g, check_tags=tags)


@pytest.mark.parametrize("idx", range(N_CASES))
def test_oracle_reproduces_call_golden(golden_dir, idx):
    run_case(os.path.join(golden_dir, "call"), idx, orc.mcall)
