"""Regenerates tests/golden/engine_products.json from the CPU oracle (oracle/ipcr_oracle.c).

Provenance: these vectors are DERIVED FROM THE RESTATED ORACLE, which is itself pinned by the
reference's literal known-answer tests (tests/test_oracle_golden.py); they are not output of the
Go reference (no Go toolchain here).  Inputs are either literals of the reference's tests or its
deterministic LCG fixture (core/engine/performance_benchmark_test.go:20-106).

    python tests/golden/make_golden.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle"))
import ipcr_oracle as O  # noqa: E402


def case(name, cfg, seq, pairs):
    prods = O.simulate_batch(cfg, seq, pairs)
    return {
        "name": name,
        "config": {"max_mm": cfg.max_mm, "terminal_window": cfg.terminal_window, "min_len": cfg.min_len,
                   "max_len": cfg.max_len, "hit_cap": cfg.hit_cap, "seed_len": cfg.seed_len, "circular": cfg.circular},
        "seq": seq if isinstance(seq, str) else None,
        "pairs": [[p.id, p.forward, p.reverse, p.min_product, p.max_product] for p in pairs],
        "products": [[p.experiment_id, p.start, p.end, p.length, p.type, p.fwd_mm, p.rev_mm, list(p.fwd_idx), list(p.rev_idx)]
                     for p in prods],
    }


def main():
    cases = []
    P, C = O.Pair, O.Config
    cases.append(case("engine_test minimal", C(), "ACGTACGTACGT", [P("test", "ACG", "ACG")]))
    cases.append(case("engine_test revcomp", C(), "TTTACGACGTAAA", [P("rev", "ACG", "TTT")]))
    cases.append(case("engine_test circular", C(circular=True), "TGACAAG", [P("p1", "AG", "TC")]))
    cases.append(case("engine_test seeded mismatch", C(max_mm=1, terminal_window=3, seed_len=12, min_len=10),
                      "CAGTACAAAAAAGGTACC", [P("seed-mm", "AAGTAC", "GGTACC")]))
    cases.append(case("oracle matrix mm1 tw3", C(max_mm=1, terminal_window=3, min_len=1, max_len=100, seed_len=12),
                      "TTTTCGTACAAAAGGTACCTTT", [P("forward_mismatch_5prime", "ACGTAC", "GGTACC")]))
    cases.append(case("oracle matrix reference N", C(max_mm=2, terminal_window=1, min_len=1, max_len=100, seed_len=12),
                      "TTTNCGTACAAAAGGTACCTTT", [P("reference_n", "ACGTAC", "GGTACC")]))
    cases.append(case("halo", C(max_mm=1, min_len=1, max_len=100, seed_len=6), "TTTACNTACAAAAGGTACCTTT",
                      [P("reference_n_inside_seed", "ACGTAC", "GGTACC")]))
    for (n, glen, mut, refn, k) in [(12, 20000, True, True, 2), (16, 20000, True, False, 1), (4, 20000, False, True, 1)]:
        seq, pairs = O.make_bench_fixture(n, glen, mut, refn)
        c = case("bench fixture pairs=%d mutate=%s refN=%s k=%d" % (n, mut, refn, k),
                 C(max_mm=k, min_len=100, max_len=240, seed_len=12), seq, pairs)
        c["fixture"] = [n, glen, mut, refn]  # regenerate the sequence with make_bench_fixture
        cases.append(c)
    with open(os.path.join(HERE, "engine_products.json"), "w") as f:
        json.dump({"provenance": "derived from the restated CPU oracle; see make_golden.py", "cases": cases}, f, indent=0)
    print("wrote", len(cases), "cases")


if __name__ == "__main__":
    main()
