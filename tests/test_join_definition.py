"""The product list stated as a definition, independently of the oracle's code.

VERDICT round 1: the oracle's join (oracle/ipcr_oracle.c) and the product path's (csrc/host.cpp) are sibling
transliterations of core/engine/engine.go:108-404, so their agreement says little about the join itself.  This file
states the same contract the other way round -- as sets built from the definition of a match and of a product
(SURVEY 8a rows 1, 4, 12, 16), ordered by a sort key instead of by nested scans with binary searches -- in plain Python
that shares no code, no data structure and no loop shape with either implementation, and compares full ordered
product lists with the oracle on random small cases (ACGT-only sequences and HitCap 0, where the per-orientation match
order is ascending by position; linear and circular, engine and pair length bounds, IUPAC primers, windows 0..5)."""
import random

import pytest

from oracle import ipcr_oracle as O

IUPAC = {"A": "A", "C": "C", "G": "G", "T": "T", "R": "AG", "Y": "CT", "S": "CG", "W": "AT", "K": "GT", "M": "AC",
         "B": "CGT", "D": "AGT", "H": "ACT", "V": "ACG", "N": "ACGT"}
COMP = {"A": "T", "C": "G", "G": "C", "T": "A", "R": "Y", "Y": "R", "S": "S", "W": "W", "K": "M", "M": "K",
        "B": "V", "V": "B", "D": "H", "H": "D", "N": "N"}


def rc(p):
    return "".join(COMP[c] for c in reversed(p))


def sites(seq, pat, k, protected):
    """{pos: mismatch positions} of every window of seq that `pat` matches with <= k mismatches, none at a protected index"""
    out = {}
    for pos in range(len(seq) - len(pat) + 1):
        mm = [j for j, c in enumerate(pat) if seq[pos + j] not in IUPAC[c]]
        if len(mm) <= k and not any(j in protected for j in mm):
            out[pos] = mm
    return out


def products_by_definition(seq, pid, fwd, rev, k, tw, min_len, max_len, circular):
    n = len(seq)
    tw = max(tw, 0)
    out = []
    for typ, first, second in (("forward", fwd, rev), ("revcomp", rev, fwd)):
        la, lb = len(first), len(second)
        left = sites(seq, first, k, set(range(max(la - tw, 0), la)))           # the primer itself: 3' window at its right end
        right = sites(seq, rc(second), k, set(range(0, min(tw, lb))))         # the partner's reverse complement: window at the left end
        rows = []
        for a, amm in left.items():
            for b, bmm in right.items():
                for wrap in ((False, True) if circular else (False,)):
                    if not wrap and b <= a:
                        continue
                    if wrap and b >= a:
                        continue
                    length = (b + lb - a) if not wrap else (n - a) + b + lb
                    if (min_len and length < min_len) or (max_len and length > max_len):
                        continue
                    rev_idx = tuple(lb - 1 - j for j in bmm)   # positions in the partner primer's own coordinates
                    rows.append(((a, wrap, -b), (pid, a, b + lb, length, typ, len(amm), len(bmm), tuple(amm), rev_idx)))
        out += [r for _, r in sorted(rows)]   # per left site ascending: linear partners right-to-left, then the wrapped ones right-to-left
    return out


@pytest.mark.parametrize("seed", range(12))
def test_product_lists_equal_the_definition(seed):
    rng = random.Random(9100 + seed)
    checked = 0
    for case in range(40):
        n = rng.choice([12, 30, 60, 120])
        seq = [rng.choice("ACGT") for _ in range(n)]
        k = rng.choice([0, 1, 2, 3])
        tw = rng.choice([0, 1, 2, 3, 5])
        pairs, opairs = [], []
        for i in range(rng.randint(1, 3)):
            def primer(L):
                s = [rng.choice("ACGT") for _ in range(L)]
                if rng.random() < 0.3:
                    s[rng.randrange(L)] = rng.choice("RYSWKMBDHVN")
                return "".join(s)
            f, r = primer(rng.randint(3, 8)), primer(rng.randint(3, 8))
            if rng.random() < 0.7:   # make it amplify something: plant the pair
                a = rng.randrange(0, max(1, n - 25))
                d = rng.randint(len(f), 20)
                site_f = [rng.choice(IUPAC[c]) for c in f]
                site_r = [rng.choice(IUPAC[c]) for c in rc(r)]
                if a + d + len(r) <= n:
                    seq[a:a + len(f)] = site_f
                    seq[a + d:a + d + len(r)] = site_r
            pmin, pmax = rng.choice([0, 0, 6]), rng.choice([0, 0, 40])
            pairs.append((f"p{i}", f, r, pmin, pmax))
            opairs.append(O.Pair(f"p{i}", f, r, pmin, pmax))
        cmin, cmax = rng.choice([0, 0, 5, 10]), rng.choice([0, 0, 30, 200])
        circular = rng.random() < 0.4
        s = "".join(seq)
        want = []
        for pid, f, r, pmin, pmax in pairs:
            want += products_by_definition(s, pid, f, r, k, tw, pmin or cmin, pmax or cmax, circular)
        cfg = O.Config(max_mm=k, terminal_window=tw, min_len=cmin, max_len=cmax, hit_cap=0, seed_len=rng.choice([0, 4, -1]),
                       circular=circular)
        got = [p.sig() for p in O.simulate_batch(cfg, s, opairs)]
        assert got == want, (s, pairs, k, tw, cmin, cmax, circular)
        checked += len(want)
    assert checked > 50
