"""Committed golden vectors (tests/golden/engine_products.json): the oracle must keep producing
them (CPU), and the HIP path must reproduce them through the C ABI (GPU)."""
import json
import os

import pytest

import ipcr_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "engine_products.json")


def load():
    with open(GOLDEN) as f:
        return json.load(f)["cases"]


def inputs(c):
    cfg = c["config"]
    if c.get("fixture"):
        n, glen, mut, refn = c["fixture"]
        seq, _ = O.make_bench_fixture(n, glen, mut, refn)
    else:
        seq = c["seq"].encode()
    return cfg, seq, c["pairs"]


def want(c):
    return [tuple(p[:7]) + (tuple(p[7]), tuple(p[8])) for p in c["products"]]


@pytest.mark.parametrize("i", range(len(load())))
def test_oracle_reproduces_golden(i):
    c = load()[i]
    cfg, seq, pairs = inputs(c)
    got = O.simulate_batch(O.Config(**cfg), seq, [O.Pair(*p) for p in pairs])
    assert [g.sig() for g in got] == want(c)
    # and the reference's own brute-force oracle agrees as a multiset (bruteforce.go:11-38)
    brute = O.simulate_bruteforce(O.Config(**cfg), seq, [O.Pair(*p) for p in pairs])
    assert sorted(b.sig() for b in brute) == sorted(want(c))


@pytest.mark.gpu
@pytest.mark.parametrize("i", range(len(load())))
def test_hip_reproduces_golden(i):
    from ipcr_amd import engine, primer
    c = load()[i]
    cfg, seq, pairs = inputs(c)
    ecfg = engine.Config(MaxMM=cfg["max_mm"], TerminalWindow=cfg["terminal_window"], MinLen=cfg["min_len"],
                         MaxLen=cfg["max_len"], HitCap=cfg["hit_cap"], SeedLen=cfg["seed_len"], Circular=cfg["circular"])
    got = engine.New(ecfg).SimulateBatch("seq", seq, [primer.Pair(*p) for p in pairs])
    assert [g.sig() for g in got] == want(c)
