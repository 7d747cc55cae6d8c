// engine_contract.cpp -- include/ipcr_hip.hpp driven the way the reference drives its engine (test program; built and run
// by tests/test_gpu_native_host.py).  Expectations: the reference's own test literals (core/engine/engine_test.go:11-148,
// core/engine/approx_seed_oracle_test.go:96-122, internal/pipeline/pipeline_engine_contract_test.go:13-28) as worked out
// in SURVEY.md appendix A; the Python suite checks the same cases against the oracle.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>
#include <tuple>
#include <vector>

#include "ipcr_hip.hpp"

using ipcr::Config;
using ipcr::Engine;
using ipcr::Pair;
using ipcr::Product;

static int failures = 0;
#define EXPECT(cond)                                                                    \
    do {                                                                                \
        if (!(cond)) { fprintf(stderr, "%s:%d: expectation failed: %s\n", __FILE__, __LINE__, #cond); ++failures; } \
    } while (0)

typedef std::tuple<long long, long long, long long, std::string, int, int, std::vector<int>, std::vector<int>> Sig;
static Sig sig(const Product &p) { return Sig(p.Start, p.End, p.Length, p.Type, p.FwdMM, p.RevMM, p.FwdMismatchIdx, p.RevMismatchIdx); }

int main() {
    if (ipcr_device_count() < 1) { fprintf(stderr, "no HIP device: the scan path has no CPU fallback\n"); return 2; }
    try {
        {   // engine_test.go:11-35: ACGTACGTACGT, A = B = ACG, exact: 6 forward + 6 revcomp, the first is (0, 12, 12)
            Engine e = Engine::New(Config{});
            std::vector<Product> got = e.SimulateBatch("s", "ACGTACGTACGT", {Pair{"x", "ACG", "ACG"}});
            EXPECT(got.size() == 12);
            EXPECT(!got.empty() && sig(got[0]) == Sig(0, 12, 12, "forward", 0, 0, {}, {}));
            EXPECT(got.size() == 12 && got[6].Type == "revcomp" && got[0].ExperimentID == "x" && got[0].SequenceID == "s");
            // :38-78: per-pair length bounds override the engine's
            got = e.SimulateBatch("s", "ACGTACGTACGT", {Pair{"x", "ACG", "ACG", 10, 12}});
            EXPECT(got.size() == 2 && sig(got[0]) == Sig(0, 12, 12, "forward", 0, 0, {}, {}) && got[1].Type == "revcomp");
            got = e.SimulateBatch("s", "ACGTACGTACGT", {Pair{"x", "ACG", "ACG", 5, 7}});
            EXPECT(got.empty());
        }
        {   // engine_test.go:81-101
            Engine e = Engine::New(Config{});
            std::vector<Product> got = e.SimulateBatch("s", "TTTACGACGTAAA", {Pair{"x", "ACG", "TTT"}});
            EXPECT(got.size() == 3);
            if (got.size() == 3) {
                EXPECT(sig(got[0]) == Sig(3, 13, 10, "forward", 0, 0, {}, {}));
                EXPECT(sig(got[1]) == Sig(6, 13, 7, "forward", 0, 0, {}, {}));
                EXPECT(sig(got[2]) == Sig(0, 10, 10, "revcomp", 0, 0, {}, {}));
            }
        }
        {   // engine_test.go:104-129: circular template, the product wraps (Start > End)
            Config c; c.Circular = true;
            std::vector<Product> lin = Engine::New(Config{}).SimulateBatch("s", "TGACAAG", {Pair{"x", "AG", "TC"}});
            std::vector<Product> cir = Engine::New(c).SimulateBatch("s", "TGACAAG", {Pair{"x", "AG", "TC"}});
            EXPECT(lin.empty());
            EXPECT(cir.size() == 1 && sig(cir[0]) == Sig(5, 3, 5, "forward", 0, 0, {}, {}));
        }
        {   // engine_test.go:131-148: one mismatch outside the 3' window
            Config c; c.MaxMM = 1; c.TerminalWindow = 3; c.MinLen = 10;
            std::vector<Product> got = Engine::New(c).SimulateBatch("s", "CAGTACAAAAAAGGTACC", {Pair{"x", "AAGTAC", "GGTACC"}});
            EXPECT(got.size() == 1 && sig(got[0]) == Sig(0, 18, 18, "forward", 1, 0, {0}, {}));
        }
        {   // approx_seed_oracle_test.go:114-122,187: k = 1, 3' window 3 -- three products with their mismatch indices
            Config c; c.MaxMM = 1; c.TerminalWindow = 3; c.MinLen = 1; c.MaxLen = 100;
            Engine e = Engine::New(c);
            ipcr::CompiledPanel cp = e.CompilePanel({Pair{"x", "ACGTAC", "GGTACC"}});
            ipcr::SimulationScratch sc = e.NewSimulationScratch(cp);
            const std::string seq = "TTTTCGTACAAAAGGTACCTTT";
            std::vector<Product> got = e.SimulateCompiledWithScratch("rec", seq, cp, sc);
            EXPECT(got.size() == 3);
            if (got.size() == 3) {
                EXPECT(sig(got[0]) == Sig(3, 19, 16, "forward", 1, 0, {0}, {}));
                EXPECT(sig(got[1]) == Sig(12, 19, 7, "forward", 1, 0, {1}, {}));
                EXPECT(sig(got[2]) == Sig(13, 20, 7, "revcomp", 0, 1, {}, {1}));
            }
            // the streaming form gives the same products in the same order ...
            std::vector<Product> streamed;
            EXPECT(e.ForEachCompiledProduct("rec", seq, cp, sc, [&](const Product &p) { streamed.push_back(p); return true; }));
            EXPECT(streamed.size() == got.size());
            for (size_t i = 0; i < streamed.size() && i < got.size(); ++i) EXPECT(sig(streamed[i]) == sig(got[i]) && streamed[i].SequenceID == "rec");
            // ... and stops at an emit error (compiled.go:141-160): false comes back, the scratch serves the next call
            int seen = 0;
            EXPECT(!e.ForEachCompiledProduct("rec", seq, cp, sc, [&](const Product &) { ++seen; return false; }));
            EXPECT(seen == 1);
            EXPECT(e.SimulateCompiledWithScratch("rec", seq, cp, sc).size() == 3);
            // pipeline.go:55-125: the panel compiled once and shared, one scratch per worker, workers on their own threads
            std::atomic<int> bad{0};
            std::vector<std::thread> th;
            for (int w = 0; w < 3; ++w)
                th.emplace_back([&, w] {
                    try {
                        ipcr::SimulationScratch mine = e.NewSimulationScratch(cp, 0);
                        for (int r = 0; r < 5; ++r) {
                            const std::string pad((size_t)(w * 7 + r), 'A');
                            std::vector<Product> g = e.SimulateCompiledWithScratch("w", pad + seq, cp, mine);
                            if (g.size() < 3) { bad.fetch_add(1); continue; }
                            // leading As may add matches of their own; the three known products are there, shifted by the pad
                            int found = 0;
                            for (const Product &p : g)
                                if ((p.Start == 3 + (long long)pad.size() && p.Length == 16) || (p.Start == 12 + (long long)pad.size() && p.Length == 7 && p.Type == "forward") ||
                                    (p.Start == 13 + (long long)pad.size() && p.Type == "revcomp")) ++found;
                            if (found != 3) bad.fetch_add(1);
                        }
                    } catch (const std::exception &) { bad.fetch_add(1); }
                });
            for (auto &t : th) t.join();
            EXPECT(bad.load() == 0);
        }
        {   // what the reference panics on is an error here (core/primer/rc.go:27-34); a negative MaxMM is refused at the boundary
            bool threw = false;
            try { (void)Engine::New(Config{}).CompilePanel({Pair{"x", "ACGU?", "ACG"}}); } catch (const ipcr::Error &er) { threw = er.status == IPCR_ERR_PRIMER; }
            EXPECT(threw);
            Config c; c.MaxMM = -1;
            threw = false;
            try { (void)Engine::New(c).CompilePanel({Pair{"x", "ACG", "ACG"}}); } catch (const ipcr::Error &er) { threw = er.status == IPCR_ERR_INVALID; }
            EXPECT(threw);
        }
    } catch (const std::exception &ex) {
        fprintf(stderr, "unexpected exception: %s\n", ex.what());
        return 1;
    }
    if (failures) { fprintf(stderr, "%d expectation(s) failed\n", failures); return 1; }
    printf("ok\n");
    return 0;
}
