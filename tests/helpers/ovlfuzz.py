"""Random read pairs for the mate-overlap tweak: CIGARs with M/=/X/I/D/N/S runs, mates placed so that they overlap
partly, fully or not at all (test infrastructure)."""
import numpy as np

from . import sam as S


class FuzzRead:
    pass


def _cigar(rng, l_qseq):
    ops, left = [], l_qseq
    if rng.random() < 0.3 and left > 4:
        n = int(rng.integers(1, 4)); ops.append((n, "S")); left -= n
    while left > 0:
        n = int(rng.integers(1, min(left, 25) + 1))
        ops.append((n, "M=X"[int(rng.integers(0, 3))] if rng.random() < 0.2 else "M")); left -= n
        if left > 0:
            k = rng.random()
            if k < 0.25:
                n = int(rng.integers(1, min(left, 4) + 1)); ops.append((n, "I")); left -= n
            elif k < 0.5:
                ops.append((int(rng.integers(1, 6)), "D"))
            elif k < 0.55:
                ops.append((int(rng.integers(5, 40)), "N"))
    if ops[-1][1] in "DN":
        ops.pop()
    if ops[-1][1] in "M=X" and rng.random() < 0.3 and ops[-1][0] > 2:
        n, op = ops.pop(); ops += [(n - 1, op), (1, "S")]
    return ops


def make_read(rng, pos, l_qseq, qmax=60):
    r = FuzzRead()
    r.pos, r.l_qseq = int(pos), int(l_qseq)
    r.cigar = _cigar(rng, l_qseq)
    r.bamcigar = np.array([n << 4 | "MIDNSHP=X".index(op) for n, op in r.cigar], dtype=np.uint32)
    r.seq = "".join("ACGTN"[i] for i in rng.choice(5, l_qseq, p=[0.3, 0.3, 0.19, 0.19, 0.02]))
    r.qual = rng.integers(0, qmax + 1, l_qseq).astype(np.int32)
    r.flag = 0
    r.end = r.pos + sum(n for n, op in r.cigar if op in "M=XDN")
    r.qname = ""
    return r


def pairs(seed, n_pairs):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n_pairs):
        la, lb = int(rng.integers(20, 120)), int(rng.integers(20, 120))
        a = make_read(rng, rng.integers(0, 1000), la, qmax=int(rng.choice([40, 93, 150])))
        # the second mate starts somewhere from before the first to past its end
        b = make_read(rng, max(0, a.pos + int(rng.integers(-30, a.end - a.pos + 20))), lb, qmax=int(rng.choice([40, 93, 150])))
        if rng.random() < 0.5:                                   # make the overlap agree often, as real mates do
            pass
        out.append((a, b))
    return out
