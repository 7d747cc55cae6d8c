// ipcr_hip.hpp -- the host side above the C ABI, in C++: the reference's simulator interface family
// (internal/pipeline/sim.go:11-39 -- Simulator, CompiledSimulator, ScratchCompiledSimulator,
// StreamingCompiledSimulator) with the same method names, argument meaning and error behaviour, over include/ipcr_hip.h.
// Header-only; link libipcr_hip.so.  The reference is Go and this image has no Go toolchain: INTEGRATION.md carries the
// cgo shim a maintainer would add; this header is the same surface for a C++ host (ipcr_amd/engine.py is the Python one).
//
//   ipcr::Engine eng = ipcr::Engine::New(cfg);                       // engine.New              core/engine/engine.go:22-30
//   ipcr::CompiledPanel cp = eng.CompilePanel(pairs);                // once, before the workers  internal/pipeline/pipeline.go:55-58
//   ipcr::SimulationScratch sc = eng.NewSimulationScratch(cp, dev);  // one per worker, never shared  pipeline.go:66-69
//   eng.ForEachCompiledProduct(id, seq, cp, sc, [&](const ipcr::Product& p) { ...; return true; });   // pipeline.go:100
//
// Threading as in the reference: a CompiledPanel is shared read-only by all workers, a SimulationScratch belongs to one
// worker.  A scratch lives on ONE device (worker i -> device i mod N; every call selects it itself: a worker thread never
// calls hipSetDevice).  emit returning false is the reference's emit error: the scan of that chunk stops and
// ForEachCompiledProduct returns false.  Coordinates are chunk-local, SequenceID is the caller's id (pipeline.go:80-89
// adds the chunk offset, dedups and fills Seq / SourceFile afterwards).  There is no CPU fallback: without a HIP device
// every scan throws ipcr::Error.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <string_view>
#include <utility>
#include <vector>

#include "ipcr_hip.h"

namespace ipcr {

struct Error : std::runtime_error {
    ipcr_status status;
    Error(ipcr_status st, const std::string &what) : std::runtime_error(what), status(st) {}
};
inline void check(ipcr_status st) {
    if (st != IPCR_OK) throw Error(st, ipcr_last_error());
}

// engine.Config -- core/engine/engine.go:10-19
struct Config {
    int MaxMM = 0, TerminalWindow = 0, MinLen = 0, MaxLen = 0, HitCap = 0, SeedLen = 0;
    bool NeedSites = false, Circular = false;
};
// primer.Pair -- core/primer/pair.go:4-10
struct Pair {
    std::string ID, Forward, Reverse;
    int MinProduct = 0, MaxProduct = 0;
};
// engine.Product -- core/engine/product.go:4-35, the fields a scan fills
struct Product {
    std::string ExperimentID, SequenceID;
    int64_t Start = 0, End = 0, Length = 0;
    std::string Type; // "forward" | "revcomp"
    int FwdMM = 0, RevMM = 0;
    std::vector<int> FwdMismatchIdx, RevMismatchIdx;
};

// engine.CompiledPanel -- core/engine/compiled.go:96-136 (+ the device tables and kernels, built per device at the first scan there)
class CompiledPanel {
public:
    CompiledPanel() = default;
    CompiledPanel(CompiledPanel &&o) noexcept : p_(o.p_), pairs_(std::move(o.pairs_)) { o.p_ = nullptr; }
    CompiledPanel &operator=(CompiledPanel &&o) noexcept {
        if (this != &o) { reset(); p_ = o.p_; pairs_ = std::move(o.pairs_); o.p_ = nullptr; }
        return *this;
    }
    CompiledPanel(const CompiledPanel &) = delete;
    CompiledPanel &operator=(const CompiledPanel &) = delete;
    ~CompiledPanel() { reset(); }
    const ipcr_panel *get() const { return p_; }
    const std::vector<Pair> &Pairs() const { return pairs_; }
    // blocks until the panel's kernels are built on every device it has been scanned on (a small panel builds them in the
    // background and its first scans take the table-driven kernel: same results)
    void WaitReady() const { check(ipcr_panel_wait_ready(const_cast<ipcr_panel *>(p_))); }

private:
    friend class Engine;
    void reset() { if (p_) ipcr_panel_destroy(p_); p_ = nullptr; }
    ipcr_panel *p_ = nullptr;
    std::vector<Pair> pairs_;
};

// engine.SimulationScratch -- core/engine/hit_collect.go:21-34: one HIP stream + staging + hit buffers, on one device
class SimulationScratch {
public:
    SimulationScratch() = default;
    SimulationScratch(SimulationScratch &&o) noexcept : s_(o.s_) { o.s_ = nullptr; }
    SimulationScratch &operator=(SimulationScratch &&o) noexcept {
        if (this != &o) { reset(); s_ = o.s_; o.s_ = nullptr; }
        return *this;
    }
    SimulationScratch(const SimulationScratch &) = delete;
    SimulationScratch &operator=(const SimulationScratch &) = delete;
    ~SimulationScratch() { reset(); }
    ipcr_scratch *get() const { return s_; }
    int Device() const { return (int)ipcr_scratch_device(s_); }
    ipcr_scan_stats Stats() const { ipcr_scan_stats st{}; check(ipcr_scratch_stats(s_, &st)); return st; }

private:
    friend class Engine;
    void reset() { if (s_) ipcr_scratch_destroy(s_); s_ = nullptr; }
    ipcr_scratch *s_ = nullptr;
};

class Engine {
public:
    static Engine New(const Config &c) { Engine e; e.cfg_ = c; return e; } // engine.New
    const Config &Cfg() const { return cfg_; }
    void SetHitCap(int n) { cfg_.HitCap = n; }                             // engine.go:29-30

    // CompiledSimulator.CompilePanel.  Throws ipcr::Error where the reference panics (a primer that is not upper-case
    // IUPAC DNA: core/primer/rc.go:27-34) and for what the device path does not take (primers > 128 nt, MaxMM > 16 or < 0).
    CompiledPanel CompilePanel(const std::vector<Pair> &pairs) const {
        std::vector<ipcr_pair> raw;
        raw.reserve(pairs.size());
        CompiledPanel cp;
        cp.pairs_ = pairs; // (the C structs below point into this copy)
        for (const Pair &p : cp.pairs_) raw.push_back(ipcr_pair{p.ID.c_str(), p.Forward.c_str(), p.Reverse.c_str(), p.MinProduct, p.MaxProduct});
        ipcr_config c{};
        c.max_mm = cfg_.MaxMM; c.terminal_window = cfg_.TerminalWindow; c.min_len = cfg_.MinLen; c.max_len = cfg_.MaxLen;
        c.hit_cap = cfg_.HitCap; c.seed_len = cfg_.SeedLen; c.circular = cfg_.Circular ? 1 : 0; c.need_sites = cfg_.NeedSites ? 1 : 0;
        check(ipcr_panel_create(&c, raw.data(), (int32_t)raw.size(), &cp.p_));
        return cp;
    }

    // ScratchCompiledSimulator.NewSimulationScratch; device < 0: the process's default device (ipcr_set_device)
    SimulationScratch NewSimulationScratch(const CompiledPanel &cp, int device = -1) const {
        SimulationScratch s;
        check(device < 0 ? ipcr_scratch_create(cp.get(), &s.s_) : ipcr_scratch_create_on(cp.get(), device, &s.s_));
        return s;
    }

    // StreamingCompiledSimulator.ForEachCompiledProduct: products in the reference's emission order (per pair: forward
    // block, then revcomp block; core/engine/engine.go:108-404).  emit(const Product&) -> bool; false stops the scan
    // (the reference's emit error) and makes this return false.
    template <class Emit>
    bool ForEachCompiledProduct(const std::string &seqID, std::string_view seq, const CompiledPanel &cp, SimulationScratch &scratch, Emit &&emit) const {
        struct Ctx { const CompiledPanel *cp; const std::string *id; Emit *emit; } ctx{&cp, &seqID, &emit};
        const ipcr_status st = ipcr_scan_chunk(cp.get(), scratch.get(), reinterpret_cast<const uint8_t *>(seq.data()), (uint64_t)seq.size(),
            [](const ipcr_product *p, void *user) -> int {
                Ctx *c = static_cast<Ctx *>(user);
                return (*c->emit)(convert(*p, *c->cp, *c->id)) ? 0 : 1;
            }, &ctx);
        if (st == IPCR_ERR_ABORTED) return false;
        check(st);
        return true;
    }

    // ScratchCompiledSimulator.SimulateCompiledWithScratch
    std::vector<Product> SimulateCompiledWithScratch(const std::string &seqID, std::string_view seq, const CompiledPanel &cp, SimulationScratch &scratch) const {
        check(ipcr_scan_chunk(cp.get(), scratch.get(), reinterpret_cast<const uint8_t *>(seq.data()), (uint64_t)seq.size(), nullptr, nullptr));
        const ipcr_product *pr = nullptr;
        int64_t n = 0;
        check(ipcr_scratch_products(scratch.get(), &pr, &n));
        std::vector<Product> out;
        out.reserve((size_t)n);
        for (int64_t i = 0; i < n; ++i) out.push_back(convert(pr[i], cp, seqID));
        return out;
    }
    // CompiledSimulator.SimulateCompiled (a scratch of its own for the call)
    std::vector<Product> SimulateCompiled(const std::string &seqID, std::string_view seq, const CompiledPanel &cp) const {
        SimulationScratch s = NewSimulationScratch(cp);
        return SimulateCompiledWithScratch(seqID, seq, cp, s);
    }
    // Simulator.SimulateBatch (compiles the panel for the call, as engine.go:33-51 does)
    std::vector<Product> SimulateBatch(const std::string &seqID, std::string_view seq, const std::vector<Pair> &pairs) const {
        CompiledPanel cp = CompilePanel(pairs);
        return SimulateCompiled(seqID, seq, cp);
    }

private:
    static Product convert(const ipcr_product &p, const CompiledPanel &cp, const std::string &seqID) {
        Product o;
        o.ExperimentID = cp.Pairs()[(size_t)p.pair].ID;
        o.SequenceID = seqID;
        o.Start = p.start; o.End = p.end; o.Length = p.length;
        o.Type = p.type == 0 ? "forward" : "revcomp";
        o.FwdMM = p.fwd_mm; o.RevMM = p.rev_mm;
        o.FwdMismatchIdx.assign(p.fwd_idx, p.fwd_idx + p.n_fwd_idx);
        o.RevMismatchIdx.assign(p.rev_idx, p.rev_idx + p.n_rev_idx);
        return o;
    }
    Config cfg_;
};

} // namespace ipcr
