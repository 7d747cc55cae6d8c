// jit.cpp -- panel-specialised k-mismatch filter for gfx950.
//
// The table-driven filter (kernels.hip) pays ~20 VALU ops per (pattern, position, word)
// because which base a pattern position wants is only known at run time.  A compiled
// ipcr panel is fixed for the whole run (CompilePanel is called once,
// internal/pipeline/pipeline.go:55-58), so -- like the reference compiling the panel into
// an Aho-Corasick automaton -- we compile it into straight-line HIP source:
//
//  * one wavefront streams one block (64 columns x 128 rows, tile_layout.h) top to bottom,
//    lane = column, 16 B/lane loads, next row-quad prefetched while the current one is used;
//  * each incoming row is expanded once into four one-hot MISMATCH planes (base != A/C/G/T,
//    invalid bases mismatch everything) kept in a register window of the last W rows;
//  * pattern position j of a window starting at row r is row r+j of the window: a register,
//    no shift.  Every pattern becomes OR-trees over named registers:
//      protected window (3' terminal window, or everything when k = 0): must be clean;
//      the other positions are cut into B blocks; a block with any mismatch is "bad";
//      at most k bad blocks may occur (generalised pigeonhole; B = #positions is exact);
//  * survivors (rare) are appended to the candidate queue for the exact verifier.
//
// The filter is sound for any block split (a window with <= k mismatches has <= k bad
// blocks); B only trades ALU work against how many false candidates reach the verifier.
#include "jit.h"

#include <hip/hip_ext.h>
#include <hip/hiprtc.h>

#include <cmath>
#include <cstdio>
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <sstream>
#include <thread>
#include <dirent.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

namespace ipcr {

struct JitFilter {
    hipModule_t module = nullptr;
    hipFunction_t fn = nullptr;
    unsigned waves_per_group = 4;
    // what the kernel was generated from: the form for small launches (a block shared by several waves) is built from it
    // when the first small launch comes -- on a thread of its own; launches until then take the kernel above
    std::vector<ipcr_dev_pattern> pats;
    std::vector<uint32_t> ids;
    bool has_ids = false, spill_only = false;
    int max_mm = 0;
    unsigned qbase = 0;
    std::string arch;
    unsigned segments = 1;            // of THIS kernel (1: one wave per block)
    JitFilter *small = nullptr;       // the segmented form, once built
    std::atomic<int> small_state{0};  // 0 not asked for, 1 being built, 2 ready, 3 cannot be built
    std::thread small_thread;
};

std::string count_code(int B, int k, int *cost = nullptr);

namespace {

struct Plan {
    int L = 0;
    std::vector<int> prot;                // protected positions
    std::vector<std::vector<int>> blocks; // unprotected positions, contiguous groups
    bool counted = false;                 // false: the unprotected part cannot exceed k mismatches
};

double match_prob(uint8_t m) { return __builtin_popcount(m & 15u) / 4.0; }

// probability that a random window passes (prot clean and <= k bad blocks)
double pass_prob(const ipcr_dev_pattern &p, const Plan &pl, int k) {
    double pp = 1.0;
    for (int j : pl.prot) pp *= match_prob(p.mask[j]);
    if (!pl.counted) return pp;
    std::vector<double> f((size_t)k + 2, 0.0); // f[t] = P(t bad blocks), last bucket = > k
    f[0] = 1.0;
    for (const auto &b : pl.blocks) {
        double clean = 1.0;
        for (int j : b) clean *= match_prob(p.mask[j]);
        std::vector<double> g((size_t)k + 2, 0.0);
        for (int t = 0; t <= k + 1; ++t) {
            g[(size_t)t] += f[(size_t)t] * clean;
            g[(size_t)std::min(t + 1, k + 1)] += f[(size_t)t] * (1.0 - clean);
        }
        f.swap(g);
    }
    double ok = 0;
    for (int t = 0; t <= k; ++t) ok += f[(size_t)t];
    return pp * ok;
}

// VALU instructions of one block's OR (jit_source: orchain emits exactly this): a position that allows one base -- or any
// base: the invalid plane -- is one input, an IUPAC code of two bases an AND of two planes that rides along in an
// and-or with another input, a code of three bases an instruction of its own; three inputs per OR.
int block_ops(const ipcr_dev_pattern &p, const std::vector<int> &blk) {
    int singles = 0, and2 = 0, ops = 0;
    for (int j : blk) {
        const int n = __builtin_popcount(p.mask[j] & 15u);
        if (n == 0) return 0;          // matches nothing: the block is a constant
        if (n == 2) ++and2;
        else { if (n == 3) ++ops; ++singles; }
    }
    for (int i = 0; i < and2; ++i) { ++ops; if (singles == 0) singles = 1; }
    while (singles > 3) { ++ops; singles -= 2; }
    return ops + (singles >= 2 ? 1 : 0);
}

Plan make_plan(const ipcr_dev_pattern &p, int k, int B) {
    Plan pl;
    pl.L = p.len;
    std::vector<int> un;
    for (int j = 0; j < p.len; ++j) {
        if (p.mask[j] & 16u) pl.prot.push_back(j);
        else un.push_back(j);
    }
    const int U = (int)un.size();
    if (U <= k) return pl; // any number of mismatches outside the protected window is fine
    pl.counted = true;
    if (B > U) B = U;
    if (B < k + 1) B = k + 1;
    pl.blocks.resize((size_t)B);
    for (int i = 0; i < U; ++i) pl.blocks[(size_t)((long)i * B / U)].push_back(un[(size_t)i]);
    // Which positions share a block is free (<= k mismatches spoil <= k blocks whatever the partition), and an OR of 2 n + 1
    // inputs costs n instructions where one of 2 n + 2 costs n + 1: a block that holds a two-base IUPAC code beside two
    // plain positions (four inputs, two instructions) gives one of them to the panel's two-position block (three plain
    // positions: still one instruction).  Greedy: move one position at a time while the instruction count falls.
    // C3 (27F / 1492R, codes M and Y): one instruction per pattern and row step fewer, 67 -> 63 in the loop.
    const bool rebalance = !getenv("IPCR_JIT_REBALANCE") || atoi(getenv("IPCR_JIT_REBALANCE")) != 0; // (read per panel: tests switch it)
    for (int round = 0; rebalance && round < 64; ++round) {
        int best_gain = 0, bx = -1, by = -1, bi = -1;
        for (int x = 0; x < B; ++x) {
            if (pl.blocks[(size_t)x].size() < 2) continue;
            const int ox = block_ops(p, pl.blocks[(size_t)x]);
            for (size_t i = 0; i < pl.blocks[(size_t)x].size(); ++i) {
                std::vector<int> xs = pl.blocks[(size_t)x];
                const int j = xs[i];
                xs.erase(xs.begin() + (long)i);
                const int nx = block_ops(p, xs);
                for (int y = 0; y < B; ++y) {
                    if (y == x) continue;
                    std::vector<int> ys = pl.blocks[(size_t)y];
                    ys.push_back(j);
                    const int gain = ox + block_ops(p, pl.blocks[(size_t)y]) - nx - block_ops(p, ys);
                    // (ties: the move that leaves the blocks' sizes closest, so that no block becomes a lone position)
                    if (gain > best_gain || (gain == best_gain && gain > 0 && ys.size() < pl.blocks[(size_t)by].size() + 1)) { best_gain = gain; bx = x; by = y; bi = (int)i; }
                }
            }
        }
        if (best_gain <= 0) break;
        const int j = pl.blocks[(size_t)bx][(size_t)bi];
        pl.blocks[(size_t)bx].erase(pl.blocks[(size_t)bx].begin() + bi);
        pl.blocks[(size_t)by].push_back(j);
    }
    for (auto &b : pl.blocks) std::sort(b.begin(), b.end());
    return pl;
}

// How many blocks?  Every row step pays the OR trees and the block counter; a window that passes
// the block test pays the exact count in the rare branch -- and so do the other 63 lanes of its wave.
// Estimated VALU instructions per row step and pattern:
//   main(B)  = sum over blocks ceil((n-1)/2) [or3 chains] + counter(B, k)
//   rare(B)  = 64 lanes * 32 strands * P(pass) * (exact counter over all unprotected positions + branch overhead)
// B = k+1 (the classic pigeonhole split) unless the primer is so degenerate that its rare branch is not rare.
// Kernels without the exact stage (many patterns: it would be emitted at W x patterns places, and hiprtc
// time grows with it) and IPCR_JIT_TARGET_PPM use a fixed selectivity target instead.
Plan choose_plan(const ipcr_dev_pattern &p, int k, bool exact_stage) {
    int U = 0;
    for (int j = 0; j < p.len; ++j)
        if (!(p.mask[j] & 16u)) ++U;
    if (U <= k) return make_plan(p, k, k + 1);
    if (getenv("IPCR_JIT_TARGET_PPM") || !exact_stage) {
        // without the exact stage every window that passes costs the wave a memory round trip at its end:
        // sweeps on MI355X put the best pass rate at ~2e-6 (k <= 2) and ~5e-6 (k = 3, where every extra
        // block costs more counter levels)
        const double target = 1e-6 * (getenv("IPCR_JIT_TARGET_PPM") ? atof(getenv("IPCR_JIT_TARGET_PPM")) : (k >= 3 ? 5.0 : 2.0));
        Plan best = make_plan(p, k, k + 1);
        for (int B = k + 2; B <= U && pass_prob(p, best, k) > target; ++B) best = make_plan(p, k, B);
        return best;
    }
    int exact_ops = 0;
    (void)count_code(U, k, &exact_ops);
    Plan best;
    double best_cost = 1e300;
    for (int B = k + 1; B <= U; ++B) {
        Plan pl = make_plan(p, k, B);
        int counter = 0;
        (void)count_code((int)pl.blocks.size(), k, &counter);
        double main_ops = counter;
        for (const auto &blk : pl.blocks) main_ops += (double)block_ops(p, blk);
        const bool exact = (int)pl.blocks.size() == U;
        const double rare = exact ? 0.0 : 2048.0 * pass_prob(p, pl, k) * (exact_ops + 24.0);
        const double cost = main_ops + rare;
        if (cost < best_cost) { best_cost = cost; best = pl; }
    }
    return best;
}

std::string plane_expr(uint8_t mask, int slot, bool &uses_n) {
    const uint8_t m = mask & 15u;
    const std::string s = std::to_string(slot);
    if (m == 15u) { uses_n = true; return "n" + s; }
    if (m == 0u) return "0xFFFFFFFFu";
    std::string e;
    int cnt = 0;
    const char *names = "acgt";
    for (int b = 0; b < 4; ++b)
        if (m & (1u << b)) {
            if (cnt++) e += " & ";
            e += std::string(1, names[b]) + s;
        }
    return cnt > 1 ? "(" + e + ")" : e;
}

} // namespace

static int env_int(const char *name, int dflt, int lo, int hi) {
    const char *v = getenv(name);
    if (!v || !*v) return dflt;
    const int x = atoi(v);
    return x < lo ? lo : (x > hi ? hi : x);
}

// "more than k of the B block flags e0..e{B-1} are set", OR-ed into f.  Two bit-sliced
// counters; the cheaper one (in VALU ops) is emitted:
//  * thermometer: u[t] = at least t bad blocks so far (one and-or per level per block);
//  * carry-save adder tree: full adders (xor, xor, bitselect) compress the flags into binary
//    weight planes; carries beyond the top bit of k only feed an overflow OR.
std::string count_code(int B, int k, int *cost) {
    std::ostringstream th;
    int th_cost = 0;
    {
        std::vector<bool> live((size_t)k + 2, false);
        for (int i = 0; i < B; ++i) {
            for (int t = std::min(i + 1, k + 1); t >= 1; --t) {
                if (t < k + 1 - (B - 1 - i)) continue; // cannot reach level k+1 any more: dead
                const std::string prev = (t == 1) ? "" : "u" + std::to_string(t - 1) + " & ";
                if (t >= 2 && !live[(size_t)t - 1]) continue;
                // (u & e) | u' as ONE v_bitop3_b32 (two issue cycles): left to itself the compiler takes v_and_or_b32 (four).
                // The top level goes straight into the verdict: f |= u_k & e
                if (t == k + 1 && t >= 2) {
                    th << "            f = ANDOR(u" << t - 1 << ", e" << i << ", f);\n";
                    live[(size_t)t] = true;
                } else if (!live[(size_t)t]) {
                    th << "            u32 u" << t << " = " << prev << "e" << i << ";\n";
                    live[(size_t)t] = true;
                } else if (t >= 2) {
                    th << "            u" << t << " = ANDOR(u" << t - 1 << ", e" << i << ", u" << t << ");\n";
                } else {
                    th << "            u" << t << " |= e" << i << ";\n";
                }
                ++th_cost;
            }
        }
        if (k + 1 < 2) { th << "            f |= u" << (k + 1) << ";\n"; ++th_cost; }
    }
    // Carry-save adder tree.  gfx950 has v_bitop3_b32 (any function of three words in one instruction), so a full
    // adder is TWO instructions (xor3, majority) and a half adder two (xor, and); outputs that do not reach the verdict
    // are neither emitted nor counted.  Example, "more than 3 of 5 flags": maj(e0,e1,e2) & maj(e3,e4,e0^e1^e2), 4 ops.
    std::ostringstream cs;
    int cs_cost = 0;
    {
        struct Node { std::string name, expr; std::vector<int> deps; bool live = false; };
        std::vector<Node> nodes;
        auto leaf_or_node = [&](const std::string &nm) { // index of the node that defines nm, -1 for an input flag
            for (size_t i = 0; i < nodes.size(); ++i)
                if (nodes[i].name == nm) return (int)i;
            return -1;
        };
        auto add = [&](const std::string &expr, const std::vector<std::string> &in) {
            Node n;
            n.name = "x" + std::to_string(nodes.size());
            n.expr = expr;
            for (const auto &i : in) { const int d = leaf_or_node(i); if (d >= 0) n.deps.push_back(d); }
            nodes.push_back(n);
            return nodes.back().name;
        };
        int wmax = 0;
        while ((2 << wmax) <= k) ++wmax; // top weight whose bit can be set in a value <= k
        std::vector<std::vector<std::string>> planes((size_t)wmax + 2);
        for (int i = 0; i < B; ++i) planes[0].push_back("e" + std::to_string(i));
        std::vector<std::string> ovf;
        for (int w = 0; w <= wmax; ++w) {
            auto &pl = planes[(size_t)w];
            while (pl.size() >= 2) {
                std::string sum, cy;
                if (pl.size() >= 3) {
                    const std::string a = pl[0], b = pl[1], c = pl[2];
                    sum = add("XOR3(" + a + ", " + b + ", " + c + ")", {a, b, c});
                    cy = add("MAJ3(" + a + ", " + b + ", " + c + ")", {a, b, c});
                    pl.erase(pl.begin(), pl.begin() + 3);
                } else {
                    const std::string a = pl[0], b = pl[1];
                    sum = add(a + " ^ " + b, {a, b});
                    cy = add(a + " & " + b, {a, b});
                    pl.erase(pl.begin(), pl.begin() + 2);
                }
                pl.push_back(sum);
                if (w == wmax) ovf.push_back(cy); else planes[(size_t)w + 1].push_back(cy);
            }
        }
        // value = sum of planes[w][0] << w  (w <= wmax); fail = overflow | value > k
        std::vector<std::string> used = ovf;
        std::string gt, eq;
        int cmp_cost = 0;
        for (int w = wmax; w >= 0; --w) {
            const std::string bw = planes[(size_t)w].empty() ? std::string("0u") : planes[(size_t)w][0];
            if (((k >> w) & 1) == 0) {
                const std::string term = eq.empty() ? bw : "(" + eq + " & " + bw + ")";
                gt = gt.empty() ? term : gt + " | " + term;
                eq = eq.empty() ? "~" + bw : eq + " & ~" + bw;
                used.push_back(bw);
                cmp_cost += 2;
            } else {
                bool lower_zero = false; // a set bit of k only matters if some lower bit of k is clear
                for (int v = w - 1; v >= 0; --v) lower_zero |= ((k >> v) & 1) == 0;
                if (!lower_zero) continue;
                eq = eq.empty() ? bw : eq + " & " + bw;
                used.push_back(bw);
                cmp_cost += 1;
            }
        }
        if (gt.empty()) cmp_cost = 0; // nothing can exceed k below the overflow weight
        for (const auto &u : (gt.empty() ? ovf : used)) {
            std::vector<int> stack;
            const int d = leaf_or_node(u);
            if (d >= 0) stack.push_back(d);
            while (!stack.empty()) {
                const int n = stack.back();
                stack.pop_back();
                if (nodes[(size_t)n].live) continue;
                nodes[(size_t)n].live = true;
                for (int dd : nodes[(size_t)n].deps) stack.push_back(dd);
            }
        }
        for (const Node &n : nodes)
            if (n.live) { cs << "            const u32 " << n.name << " = " << n.expr << ";\n"; ++cs_cost; }
        cs << "            f |= ";
        bool first = true;
        for (const auto &o : ovf) { cs << (first ? "" : " | ") << o; first = false; }
        if (!gt.empty()) { cs << (first ? "" : " | ") << gt; first = false; }
        if (first) cs << "0u";
        cs << ";\n";
        cs_cost += cmp_cost + (int)(ovf.size() + 1) / 2; // the final ORs, three inputs per instruction
    }
    const int force = env_int("IPCR_JIT_COUNTER", 0, 0, 2); // 1 = thermometer, 2 = adder tree
    const bool use_cs = force == 2 || (force == 0 && cs_cost < th_cost);
    if (cost) *cost = use_cs ? cs_cost : th_cost;
    return use_cs ? cs.str() : th.str();
}

std::string jit_source(const std::vector<ipcr_dev_pattern> &full_pats, int k, unsigned qbase,
                       const std::vector<uint32_t> *ids, bool spill_only, int segments) {
    if (full_pats.empty() || full_pats.size() > 48 || (ids && ids->size() != full_pats.size())) return "";
    // segments > 1: the form for SMALL launches (a 4 Mb chunk is 16 blocks: 16 waves on a device with room for thousands, each
    // streaming its block's 37 row quads one behind the other -- ~25 us whatever the launch size).  A block is then shared by
    // `segments` waves: wave s tests the windows that END in iterations [e0, e1) of the rolled loop and runs the iteration in
    // front of them only to fill its register window.  Every wave loads the head quads of its block into the stash itself
    // (only the wave that walks iteration 0 has streamed them).  The rolled loop's 10 % in the sweep do not matter here -- a
    // small launch is latency, not bandwidth -- and it builds in two thirds of the time.
    const int SEG = segments < 1 ? 1 : segments;
    const bool seg = SEG > 1;
    auto pid = [&](size_t q) { return ids ? (unsigned)(*ids)[q] : qbase + (unsigned)q; }; // index in the panel's device table
    // The register window is the kernel's budget (W rows x 4 planes), and the filter only has to be SOUND: a window
    // with <= k mismatches has <= k mismatches at any subset of its positions.  So a pattern longer than LF is
    // filtered by its LF positions next to the protected end (the others are left to the exact verifier, which reads
    // the tiles): 20-row windows leave registers for two row-quads of prefetch whatever the primer lengths.
    // off = positions dropped at the window's start: the filter then sees the window `off` rows late.
    const int LF = env_int("IPCR_JIT_FILTER_LEN", 20, 8, 32);
    std::vector<ipcr_dev_pattern> pats = full_pats;
    std::vector<int> offs(pats.size(), 0);
    int Lmax = 0;
    bool long_pattern = false; // a primer of 33..128 nt: its survivors go to the stand-alone verifier (the wave's own takes <= 32 nt)
    for (size_t q = 0; q < pats.size(); ++q) {
        ipcr_dev_pattern &p = pats[q];
        if (p.len == 0 || p.len > 128) return "";
        long_pattern |= p.len > 32;
        if ((int)p.len > LF) {
            const int drop = (int)p.len - LF;
            if (p.mask[p.len - 1] & 16u) { // protected 3' window at the right end (or everything protected): keep the right end
                for (int j = 0; j < LF; ++j) p.mask[j] = p.mask[j + drop];
                offs[q] = drop;
            }
            for (int j = LF; j < (int)p.len; ++j) p.mask[j] = 0;
            p.len = (uint16_t)LF;
        }
        Lmax = std::max<int>(Lmax, p.len);
    }
    // IPCR_JIT_MERGE=1: ONE rare-branch test per row quad instead of one per row.  The rare branch (some window passed the
    // block test: exact count, push) reads the window's rows, and the row after a window's last one overwrites its first --
    // unless the register window has spare slots: with three of them a quad's four rows can all be expanded and tested
    // before any of their rare branches runs.  W = 24 slots for windows of 17..20 rows (20 more registers, no spill at
    // two waves per SIMD) buys 15 fewer v_cmp + s_cbranch_vccz pairs per 20 rows.  Needs the peeled loop (no row guards).
    // MEASURED (3 Gb): C3 0.1958 / 0.1994 ms against 0.1970 / 0.2002 without, C2 0.1806 / 0.1876 against 0.1830 / 0.1815 --
    // inside the run-to-run spread, for 15 % more generated source (a 24-row loop body): off by default, parity-tested.
    const bool roll = seg || env_int("IPCR_JIT_ROLL", 0, 0, 1) != 0;
    const bool want_merge = !roll && env_int("IPCR_JIT_PEEL", 1, 0, 1) != 0 && env_int("IPCR_JIT_MERGE", 0, 0, 1) != 0 && Lmax <= 20;
    const int W = (Lmax + 3 + (want_merge ? 3 : 0)) / 4 * 4; // window rows, multiple of the row-quad (merge: at least three spare)
    // tuning knobs; defaults = best of the sweeps on MI355X (tools/sweep_jit.py): windows up to
    // 20 rows leave registers for two quads of prefetch, wider ones spill unless it is one
    const int D = env_int("IPCR_JIT_DEPTH", Lmax <= 20 ? 2 : 1, 1, 4); // row-quads prefetched ahead
    const int WPS = env_int("IPCR_JIT_WAVES", 2, 1, 4);                // __launch_bounds__ waves per SIMD
    const int WPG = env_int("IPCR_JIT_WG", Lmax <= 20 ? 2 : 4, 1, 4);  // waves per workgroup
    const int QPI = W / 4;                   // quads per unrolled main-loop iteration
    const int LM1 = Lmax - 1;                // rows of the next strand a window can reach
    const int QTOTAL = (128 + LM1 + 3) / 4;  // quads streamed: 32 of the strand + the wrap rows
    const int QM = ((32 - D) / QPI) * QPI;   // quads done by the rolled main loop (its prefetch stays < 32)
    const int NFULL = QM / QPI;
    const int QW = (LM1 + 3) / 4;            // head quads stashed for the wrap phase
    // survivor words a wave keeps for its own verify pass; 0 = every survivor spills to the global queue and the host
    // runs the stand-alone verifier (kernels with a primer longer than 32 nt)
    const int LIST_CAP = (long_pattern || spill_only) ? 0 : env_int("IPCR_JIT_LIST", 48, 1, 64);
    const int CAND_CAP = env_int("IPCR_JIT_CANDS", 128, 2, 1024); // candidate windows verified two per load round

    // exact count in the rare branch (see row_code): emitted once per (row slot, pattern), so only for small kernels
    const bool exact_stage = env_int("IPCR_JIT_EXACT", 1, 0, 1) != 0 && (int)pats.size() * W <= env_int("IPCR_JIT_EXACT_SITES", 160, 0, 100000);
    bool uses_n = false;
    std::vector<Plan> plans;
    for (const auto &p : pats) plans.push_back(choose_plan(p, k, exact_stage));
    // evaluation of every pattern for the window that starts at slot sr
    auto eval_code = [&](int sr, const std::string &sfx) {
        std::ostringstream o;
        for (size_t q = 0; q < pats.size(); ++q) {
            const Plan &pl = plans[q];
            const auto &p = pats[q];
            // OR over the mismatch planes of a set of positions, written as explicit three-input instructions
            // (v_bitop3_b32 / v_or3_b32): left to itself the compiler shares sub-expressions between the OR trees and
            // the adders behind them and ends up with two-input chains, a third more instructions.  A position whose
            // IUPAC code allows two or three bases is the AND of that many planes.
            int tmpn = 0;
            auto orchain = [&](const std::vector<int> &js) {
                std::vector<std::string> singles;
                std::vector<std::vector<std::string>> ands;
                for (int j : js) {
                    const uint8_t m = p.mask[j] & 15u;
                    const std::string sl = std::to_string((sr + j) % W);
                    if (m == 15u) { uses_n = true; singles.push_back("n" + sl); continue; }
                    if (m == 0u) return std::string("0xFFFFFFFFu"); // matches nothing
                    std::vector<std::string> t;
                    for (int b = 0; b < 4; ++b)
                        if (m & (1u << b)) t.push_back(std::string(1, "acgt"[b]) + sl);
                    if (t.size() == 1) singles.push_back(t[0]); else ands.push_back(t);
                }
                auto tmp = [&](const std::string &expr) {
                    const std::string nm = "o" + std::to_string(q) + "_" + std::to_string(tmpn++);
                    o << "            const u32 " << nm << " = " << expr << ";\n";
                    return nm;
                };
                for (const auto &t : ands) {
                    if (t.size() == 3) singles.push_back(tmp("AND3(" + t[0] + ", " + t[1] + ", " + t[2] + ")"));
                    else if (!singles.empty()) { const std::string x = singles.back(); singles.pop_back(); singles.push_back(tmp("ANDOR(" + t[0] + ", " + t[1] + ", " + x + ")")); }
                    else singles.push_back(tmp(t[0] + " & " + t[1]));
                }
                while (singles.size() > 3) {
                    const std::string a = singles[0], b = singles[1], c = singles[2];
                    singles.erase(singles.begin(), singles.begin() + 3);
                    singles.push_back(tmp("OR3(" + a + ", " + b + ", " + c + ")"));
                }
                if (singles.empty()) return std::string("0u");
                if (singles.size() == 1) return singles[0];
                if (singles.size() == 2) return singles[0] + " | " + singles[1];
                return "OR3(" + singles[0] + ", " + singles[1] + ", " + singles[2] + ")";
            };
            o << "          { // pattern " << q << ": len " << pl.L << ", " << pl.prot.size() << " protected, "
              << pl.blocks.size() << " blocks\n";
            { const std::string e = orchain(pl.prot); o << "            u32 f = " << e << ";\n"; }
            if (pl.counted) {
                const int B = (int)pl.blocks.size();
                for (int i = 0; i < B; ++i) { const std::string e = orchain(pl.blocks[(size_t)i]); o << "            const u32 e" << i << " = " << e << ";\n"; }
                o << count_code(B, k);
            }
            o << "            f" << q << sfx << " = f;\n          }\n";
        }
        return o.str();
    };
    // The rare branch of a row (some strand of some pattern passed the block test): per pattern the exact count, where the
    // plan has one -- that code names the window's registers, so it exists per (row slot, pattern) -- and then ONE push
    // site for all patterns: a lane walks the patterns it still has bits for (IPCR_JIT_ONE_PUSH=0: a push per pattern, the
    // form until round 4 -- three quarters of the generated instructions were those 196 inlined pushes: hiprtc's time is
    // proportional to the instruction count, 16 300 -> 11 800 for C2).
    const bool one_push = env_int("IPCR_JIT_ONE_PUSH", 1, 0, 1) != 0;
    auto rare_code = [&](int sr0, const std::string &cs) {
        std::ostringstream b;
        for (size_t q = 0; q < pats.size(); ++q) {
            const Plan &pl = plans[q];
            size_t U = 0;
            for (const auto &blk : pl.blocks) U += blk.size();
            const bool exact_already = !pl.counted || pl.blocks.size() == U;
            const bool refine = !exact_already && exact_stage;
            if (one_push && !refine) { b << "            const u32 w" << q << " = ~f" << q << cs << ";\n"; continue; }
            b << "            u32 w" << q << " = 0u;\n";
            b << "            if (f" << q << cs << " != 0xFFFFFFFFu) {\n";
            b << "              u32 f = f" << q << cs << ";\n";
            if (refine) {
                b << "              {\n";
                int e = 0;
                for (const auto &blk : pl.blocks)
                    for (int j : blk) b << "            const u32 e" << e++ << " = " << plane_expr(pats[q].mask[j], (sr0 + j) % W, uses_n) << ";\n";
                b << count_code((int)U, k);
                b << "              }\n";
            }
            if (one_push) b << "              w" << q << " = ~f;\n";
            else if (offs[q] == 0)
                b << "              if (f != 0xFFFFFFFFu) push(" << pid(q) << "ull, pos, ~f, lcnt, lkey, lbits, queue, qcap, qcount, counts);\n";
            else // the window starts offs[q] rows before the filtered part: in the previous strand (= the previous bit) when that crosses row 0
                b << "              if (f != 0xFFFFFFFFu) { u64 wp = pos; u32 wm = ~f; if (wp >= " << offs[q] << "ull) wp -= " << offs[q]
                  << "ull; else { wp += " << 128 - offs[q] << "ull; wm >>= 1; } if (wm) push(" << pid(q)
                  << "ull, wp, wm, lcnt, lkey, lbits, queue, qcap, qcount, counts); }\n";
            b << "            }\n";
        }
        if (one_push) {
            b << "            u32 pm = 0u";
            for (size_t q = 0; q < pats.size(); ++q) b << " | (w" << q << " ? " << (1u << q) << "u : 0u)";
            b << ";\n";
            b << "            while (pm) { // (lane-divergent: a lane walks the patterns it has surviving strands for)\n"
                 "              const u32 q = (u32)__builtin_ctz(pm); pm &= pm - 1u;\n"
                 "              u32 wm = w0;\n";
            for (size_t q = 1; q < pats.size(); ++q) b << "              if (q == " << q << "u) wm = w" << q << ";\n";
            // pattern id | rows dropped at the window's start << 16: constants, NOT a table in memory -- a load inside this loop makes
            // the compiler lose count of the tile loads in flight where the rare path joins the main one (vmcnt(3) for vmcnt(6))
            b << "              u32 info = " << (pid(0) | ((unsigned)offs[0] << 16)) << "u;\n";
            for (size_t q = 1; q < pats.size(); ++q) b << "              if (q == " << q << "u) info = " << (pid(q) | ((unsigned)offs[q] << 16)) << "u;\n";
            b << "              const u32 off = info >> 16;\n"
                 "              u64 wp = pos;\n"
                 "              // the window starts `off` rows before the filtered part: in the previous strand (= the previous bit) when that crosses row 0\n"
                 "              if (off) { if (wp >= (u64)off) wp -= (u64)off; else { wp += (u64)(128u - off); wm >>= 1; } }\n"
                 "              if (wm) push((u64)(info & 0xFFFFu), wp, wm, lcnt, lkey, lbits, queue, qcap, qcount, counts);\n"
                 "            }\n";
        }
        return b.str();
    };
    // one row step: expand the row into mismatch planes at `slot`, then (if a window ends
    // here) evaluate all patterns for the window starting LM1 rows earlier
    auto row_code = [&](int slot, char comp, const std::string &xexpr, const std::string &guard) {
        std::ostringstream b;
        const std::string sl = std::to_string(slot);
        b << "      { // window slot " << slot << "\n";
        b << "        const u32 lo = clo." << comp << ", hi = chi." << comp << ", iv = civ." << comp << ";\n";
        b << "        a" << sl << " = OR3(lo, hi, iv); c" << sl << " = __builtin_amdgcn_bitop3_b32(lo, hi, iv, 0xEF); g" << sl
          << " = __builtin_amdgcn_bitop3_b32(lo, hi, iv, 0xFB); t" << sl << " = __builtin_amdgcn_bitop3_b32(lo, hi, iv, 0xBF); n" << sl << " = iv;\n";
        if (guard == "never") { b << "      }\n"; return b.str(); }
        if (!guard.empty()) b << "        if (" << guard << ")\n";
        b << "        {\n          u32 ";
        for (size_t q = 0; q < pats.size(); ++q) b << (q ? ", f" : "f") << q;
        b << ";\n";
        b << eval_code(((slot - LM1) % W + W) % W, "");
        b << "          u32 all = f0";
        for (size_t q = 1; q < pats.size(); ++q) b << " & f" << q;
        b << ";\n";
        // rare: some window passed the block test.  Its rows are still in the register window, so the exact
        // mismatch count of the 32 strands is taken right here (bit-sliced adder over the unprotected positions,
        // ~2 ops per position); only windows with <= k mismatches go on to the list, and those are real hits
        // unless they cross a record end.  The block test can therefore be coarse (few blocks, few ops per row).
        b << "          if (__builtin_expect(all != 0xFFFFFFFFu, 0)) {\n";
        b << "            const u64 pos = posbase + (u64)(" << xexpr << " - " << LM1 << "u);\n";
        const int sr0 = ((slot - LM1) % W + W) % W;
        b << rare_code(sr0, "");
        b << "          }\n        }\n      }\n";
        return b.str();
    };
    // ---- the merged form: a quad's rows are expanded and tested first (row_head: f<q>_<c>, all_<c>), then ONE branch
    // covers the rare work of all four (row_rare)
    auto row_head = [&](int slot, char comp, int c, bool ev) {
        std::ostringstream b;
        const std::string sl = std::to_string(slot), cs = "_" + std::to_string(c);
        b << "      { // window slot " << slot << "\n";
        b << "        const u32 lo = clo." << comp << ", hi = chi." << comp << ", iv = civ." << comp << ";\n";
        b << "        a" << sl << " = OR3(lo, hi, iv); c" << sl << " = __builtin_amdgcn_bitop3_b32(lo, hi, iv, 0xEF); g" << sl
          << " = __builtin_amdgcn_bitop3_b32(lo, hi, iv, 0xFB); t" << sl << " = __builtin_amdgcn_bitop3_b32(lo, hi, iv, 0xBF); n" << sl << " = iv;\n";
        if (ev) {
            b << eval_code(((slot - LM1) % W + W) % W, cs);
            b << "          all" << cs << " = f0" << cs;
            for (size_t q = 1; q < pats.size(); ++q) b << " & f" << q << cs;
            b << ";\n";
        }
        b << "      }\n";
        return b.str();
    };
    auto row_rare = [&](int slot, int c, const std::string &xexpr) {
        std::ostringstream b;
        const std::string cs = "_" + std::to_string(c);
        b << "          if (all" << cs << " != 0xFFFFFFFFu) {\n";
        b << "            const u64 pos = posbase + (u64)(" << xexpr << " - " << LM1 << "u);\n";
        const int sr0 = ((slot - LM1) % W + W) % W;
        b << rare_code(sr0, cs);
        b << "          }\n";
        return b.str();
    };
    // the four rows of a quad: slot0 = the first row's slot, xs = each row's number as an expression, ev = is a window of this strand tested there
    auto quad_rows_merged = [&](int slot0, const std::string xs[4], const bool ev[4]) {
        std::ostringstream b;
        int nev = 0;
        for (int c = 0; c < 4; ++c) nev += ev[c] ? 1 : 0;
        if (nev) {
            b << "      u32 ";
            bool first = true;
            for (int c = 0; c < 4; ++c) {
                if (!ev[c]) continue;
                for (size_t q = 0; q < pats.size(); ++q) { b << (first ? "" : ", ") << "f" << q << "_" << c; first = false; }
                b << ", all_" << c;
            }
            b << ";\n";
        }
        for (int c = 0; c < 4; ++c) b << row_head(slot0 + c, "xyzw"[c], c, ev[c]);
        if (nev) {
            b << "      {\n        u32 any = ";
            bool first = true;
            for (int c = 0; c < 4; ++c) if (ev[c]) { b << (first ? "" : " & ") << "all_" << c; first = false; }
            b << ";\n        if (__builtin_expect(any != 0xFFFFFFFFu, 0)) {\n";
            for (int c = 0; c < 4; ++c) if (ev[c]) b << row_rare(slot0 + c, c, xs[c]);
            b << "        }\n      }\n";
        }
        return b.str();
    };
    auto load_normal = [&](const std::string &q) {
        const std::string d = std::to_string(D);
        if (env_int("IPCR_JIT_NT", 1, 0, 1)) // every tile byte is read once: non-temporal loads
            return "p" + d + "lo = __builtin_nontemporal_load(own + (" + q + ") * 192u); p" + d + "hi = __builtin_nontemporal_load(own + (" + q +
                   ") * 192u + 64u); p" + d + "iv = __builtin_nontemporal_load(own + (" + q + ") * 192u + 128u);";
        return "p" + d + "lo = own[(" + q + ") * 192u]; p" + d + "hi = own[(" + q + ") * 192u + 64u]; p" + d + "iv = own[(" + q + ") * 192u + 128u];";
    };
    // rows past the strand end belong to the next strand: the same words shifted down one bit,
    // bit 31 coming from the neighbour column (lane + 1, or lane 0 of the next block).  The head
    // quads were stashed in LDS when first streamed (each wave its own slice, so no barrier), so
    // nothing is fetched from HBM twice and the neighbour word is just the next lane's LDS slot.
    // Lane 63's neighbour is column 0 of the NEXT block.  IPCR_JIT_NEIGHBOUR=1 (default): its head words (QW quads x 3
    // planes) are loaded at the kernel's start, one by each of the first lanes, and kept as a 65th column of the stash --
    // the wrap quads then read LDS only.  0: lane 63 loads them where they are needed, a divergent branch with three global
    // loads per wrap quad whose round trip the wave waits for.
    const bool nb_lds = env_int("IPCR_JIT_NEIGHBOUR", 1, 0, 1) != 0;
    auto load_wrap = [&](int kq) {
        std::ostringstream b;
        const std::string nl = nb_lds ? "lane + 1u" : "(lane + 1u) & 63u";
        b << "{ v4 nlo = st[" << kq * 3 << "][" << nl << "], nhi = st[" << kq * 3 + 1 << "][" << nl << "], niv = st["
          << kq * 3 + 2 << "][" << nl << "];\n";
        if (!nb_lds)
        b << "        if (lane == 63u) { nlo = nblk[" << kq * 192 << "]; nhi = nblk[" << kq * 192 + 64 << "]; niv = nblk["
          << kq * 192 + 128 << "]; }\n";
        b << "        p" << D << "lo = (st[" << kq * 3 << "][lane] >> 1) | (nlo << 31); p" << D << "hi = (st[" << kq * 3 + 1
          << "][lane] >> 1) | (nhi << 31); p" << D << "iv = (st[" << kq * 3 + 2 << "][lane] >> 1) | (niv << 31); }";
        return b.str();
    };

    std::string shift; // advance the prefetch ring by one quad
    for (int i = 1; i < D; ++i) {
        const std::string a = std::to_string(i), b2 = std::to_string(i + 1);
        shift += "p" + a + "lo = p" + b2 + "lo; p" + a + "hi = p" + b2 + "hi; p" + a + "iv = p" + b2 + "iv; ";
    }
    // IPCR_JIT_ROLL=1: ONE loop over all the quads -- the strand's 32 and the wrap rows behind them; what a quad loads ahead
    // (tiles, or the stashed head quads shifted by one strand) and whether a row's windows are evaluated are wave-uniform
    // branches on the iteration number.  By default the loop stops where its prefetch reaches the strand's end and quads
    // [QM, QTOTAL) -- 7 of 37 for a 20-row window -- are emitted a second time as a static epilogue: 57 % of the generated
    // source, and hiprtc's time is proportional to it.  MEASURED (C2 / C3, 3 Gb): the rolled form builds in 0.46 / 0.50 s
    // instead of 0.82 / 0.86 -- and sweeps 10-13 % slower (0.203 against 0.183 ms, 0.226 against 0.200): with the tile
    // loads inside a branch the compiler no longer knows how many loads are in flight where the paths meet and waits with
    // vmcnt(0..2) where the straight-line loop waits with vmcnt(6..8) -- the two quads of prefetch are gone.  A kernel
    // that is built once per panel and swept for as long as the run lasts keeps the fast loop; the build is off the
    // critical path anyway (the first scans take the table-driven kernel).  Parity-tested (test_random_differential).
    const int NIT = roll ? (QTOTAL + QPI - 1) / QPI : NFULL;
    if (seg) { // every wave's first prefetch must lie inside the strand (quads < 32), and every wave must have an iteration to test
        if (NIT < SEG) return "";
        const int f0_last = ((SEG - 1) * NIT) / SEG - 1;
        if (f0_last * QPI + D > 32) return "";
    }
    auto load_wrap_dyn = [&](const std::string &kq) { // load_wrap for a quad number known at run time (wave-uniform)
        std::ostringstream b;
        b << "{ const u32 kq = " << kq << ";\n";
        const std::string nl = nb_lds ? "lane + 1u" : "(lane + 1u) & 63u";
        b << "        v4 nlo = st[kq * 3u][" << nl << "], nhi = st[kq * 3u + 1u][" << nl << "], niv = st[kq * 3u + 2u][" << nl << "];\n";
        if (!nb_lds)
        b << "        if (lane == 63u) { nlo = nblk[kq * 192u]; nhi = nblk[kq * 192u + 64u]; niv = nblk[kq * 192u + 128u]; }\n";
        b << "        p" << D << "lo = (st[kq * 3u][lane] >> 1) | (nlo << 31); p" << D << "hi = (st[kq * 3u + 1u][lane] >> 1) | (nhi << 31); p" << D
          << "iv = (st[kq * 3u + 2u][lane] >> 1) | (niv << 31); }";
        return b.str();
    };
    // IPCR_JIT_PEEL=1: iteration 0 of the main loop -- the rows that only fill the window, their windows would start
    // before the strand -- is emitted by itself (a row there is its four expansion instructions), and the loop that
    // follows has no "it > 0" test in front of its first LM1 rows and no "it == 0" in front of the stash stores: 19 + 5
    // wave-uniform branches fewer per 20 rows, and straight-line code for the scheduler.
    const bool peel = !roll && NFULL >= 2 && env_int("IPCR_JIT_PEEL", 1, 0, 1) != 0;
    const bool merge = want_merge && peel && W - Lmax >= 3;
    std::ostringstream pro; // iteration 0 by itself
    for (int u4 = 0; u4 < QPI && peel; ++u4) {
        pro << "  { // quad " << u4 << "\n";
        pro << "      const v4 clo = p1lo, chi = p1hi, civ = p1iv;\n";
        pro << "      " << shift << "\n";
        pro << "      " << load_normal(std::to_string(u4 + D) + "u") << "\n";
        if (u4 < QW)
            pro << "      st[" << u4 * 3 << "][lane] = clo; st[" << u4 * 3 + 1 << "][lane] = chi; st[" << u4 * 3 + 2 << "][lane] = civ;\n";
        if (merge) {
            std::string xs[4];
            bool ev[4];
            for (int c = 0; c < 4; ++c) { xs[c] = std::to_string(u4 * 4 + c) + "u"; ev[c] = u4 * 4 + c >= LM1; }
            pro << quad_rows_merged(u4 * 4, xs, ev);
        } else
        for (int c = 0; c < 4; ++c) {
            const int step = u4 * 4 + c;
            pro << row_code(step, "xyzw"[c], std::to_string(step) + "u", step < LM1 ? "never" : "");
        }
        pro << "  }\n";
    }
    std::ostringstream body; // rolled main loop: quads [0, QM), or all of them
    for (int u4 = 0; u4 < QPI; ++u4) {
        body << "    { // quad " << u4 << " of the iteration\n";
        body << "      const u32 qi = it * " << QPI << "u + " << u4 << "u;\n";
        const bool cut = roll && (NIT - 1) * QPI + u4 >= QTOTAL; // the last iteration has no such quad
        if (cut) body << "      if (it < " << NIT - 1 << "u) {\n";
        body << "      const v4 clo = p1lo, chi = p1hi, civ = p1iv;\n";
        body << "      " << shift << "\n";
        if (roll) {
            body << "      if (qi + " << D << "u < 32u) { " << load_normal("qi + " + std::to_string(D) + "u") << " }\n";
            body << "      else if (qi + " << D << "u < " << QTOTAL << "u) " << load_wrap_dyn("qi + " + std::to_string(D) + "u - 32u") << "\n";
        } else
            body << "      " << load_normal("qi + " + std::to_string(D) + "u") << "\n";
        if (u4 < QW && !peel && !seg)
            body << "      if (it == 0u) { st[" << u4 * 3 << "][lane] = clo; st[" << u4 * 3 + 1 << "][lane] = chi; st[" << u4 * 3 + 2
                 << "][lane] = civ; }\n";
        if (merge) {
            std::string xs[4];
            bool ev[4];
            for (int c = 0; c < 4; ++c) { xs[c] = "(qi * 4u + " + std::to_string(c) + "u)"; ev[c] = true; }
            body << quad_rows_merged(u4 * 4, xs, ev);
        } else
        for (int c = 0; c < 4; ++c) {
            const int step = u4 * 4 + c;
            std::string guard = step < LM1 && !peel ? "it > 0u" : "";
            if (seg) guard += std::string(guard.empty() ? "" : " && ") + "it >= seg_e0"; // (the iteration in front of a wave's own only fills the window)
            if (roll) { // row x = it * W + step ends a window of this strand iff LM1 <= x < 128 + LM1
                const int it_max = (128 + LM1 - 1 - step) / W;
                if (it_max < 0) guard = "never";
                else if (it_max < NIT - 1) guard += std::string(guard.empty() ? "" : " && ") + "it < " + std::to_string(it_max + 1) + "u";
            }
            body << row_code(step, "xyzw"[c], "(qi * 4u + " + std::to_string(c) + "u)", guard);
        }
        if (cut) body << "      }\n";
        body << "    }\n";
    }
    std::ostringstream epi; // static epilogue: quads [QM, QTOTAL)
    for (int qi = QM; qi < QTOTAL && !roll; ++qi) {
        epi << "  { // quad " << qi << "\n";
        epi << "      const v4 clo = p1lo, chi = p1hi, civ = p1iv;\n";
        epi << "      " << shift << "\n";
        const int qn = qi + D;
        if (qn < 32) epi << "      " << load_normal(std::to_string(qn) + "u") << "\n";
        else if (qn < QTOTAL) epi << "      " << load_wrap(qn - 32) << "\n";
        if (merge) {
            std::string xs[4];
            bool ev[4];
            for (int c = 0; c < 4; ++c) { const int x = qi * 4 + c; xs[c] = std::to_string(x) + "u"; ev[c] = x >= LM1 && x < 128 + LM1; }
            epi << quad_rows_merged((qi % QPI) * 4, xs, ev);
        } else
        for (int c = 0; c < 4; ++c) {
            const int x = qi * 4 + c;
            const int slot = (qi % QPI) * 4 + c;
            const bool ev = x >= LM1 && x < 128 + LM1;
            epi << row_code(slot, "xyzw"[c], std::to_string(x) + "u", ev ? "" : "never");
        }
        epi << "  }\n";
    }

    std::ostringstream s;
    s << "// generated by ipcr_amd/csrc/jit.cpp for a panel of " << pats.size() << " patterns, k = " << k << "\n";
    s << "#ifndef __HIPCC_RTC__\n#include <hip/hip_runtime.h>\n#endif\n";
    s << "typedef unsigned int u32;\ntypedef unsigned long long u64;\n";
    s << "typedef u32 v4 __attribute__((ext_vector_type(4)));\n";
    s << "struct qent { u64 key; u32 bits; u32 pad; };\n";
    s << "struct hitrec { u64 pos; u32 record; u32 pattern; u64 m0, m1; };\n";
    // three-input logic as single instructions (v_bitop3_b32; truth table over the masks a = 0xF0, b = 0xCC, c = 0xAA)
    s << "#define OR3(a, b, c) __builtin_amdgcn_bitop3_b32(a, b, c, 0xFE)\n"
         "#define AND3(a, b, c) __builtin_amdgcn_bitop3_b32(a, b, c, 0x80)\n"
         "#define ANDOR(a, b, c) __builtin_amdgcn_bitop3_b32(a, b, c, 0xEA) /* (a & b) | c */\n"
         "#define XOR3(a, b, c) __builtin_amdgcn_bitop3_b32(a, b, c, 0x96)\n"
         "#define MAJ3(a, b, c) __builtin_amdgcn_bitop3_b32(a, b, c, 0xE8)\n";
    s << "struct dpat { unsigned short len, seed_off, seed_len, reserved; u32 global_id; unsigned char mask[128]; };\n";
    s << "#define LIST_CAP " << LIST_CAP << "u\n";
    // position -> (block, row, lane, bit) of the strand-major tiles (tile_layout.h)
    s << "__device__ __forceinline__ u32 base_bits(const u32* __restrict__ planes, u64 P) {\n"
         "  const u64 strand = P >> 7, col = strand >> 5;\n"
         "  const u32 row = (u32)P & 127u, bit = (u32)strand & 31u;\n"
         "  const u64 w = (((((col >> 6) * 32u + (row >> 2)) * 3u) * 64u + (col & 63u)) << 2) + (row & 3u);\n"
         "  return ((planes[w] >> bit) & 1u) | (((planes[w + 256u] >> bit) & 1u) << 1) | (((planes[w + 512u] >> bit) & 1u) << 2);\n"
         "}\n"
         "__device__ __forceinline__ u32 rst_bit(const u32* __restrict__ rst, u64 P) {\n"
         "  const u64 strand = P >> 7, col = strand >> 5;\n"
         "  const u32 row = (u32)P & 127u, bit = (u32)strand & 31u;\n"
         "  return (rst[((((col >> 6) * 32u + (row >> 2)) * 64u + (col & 63u)) << 2) + (row & 3u)] >> bit) & 1u;\n"
         "}\n";
    // A surviving word goes to this wave's list in LDS and is verified by the wave itself once its
    // block is done; only when the list is full (dense matches: low-complexity genome against a
    // low-complexity primer) does it spill to the global queue of the stand-alone verifier, and
    // counts[0] tells the host that that kernel has to run.
    // qcount: this wave's shard counter, queue: its segment, qcap: capacity of one segment
    s << "__device__ __forceinline__ void push(u64 q, u64 pos, u32 bits, u32* lcnt, u64* lkey, u32* lbits,\n"
         "    qent* queue, u64 qcap, u64* qcount, u64* counts) {\n"
         "  const u32 i = atomicAdd(lcnt, 1u);\n"
         "  if (i < LIST_CAP) { lkey[i] = (q << 48) | pos; lbits[i] = bits; return; }\n"
         "  atomicAdd(counts, 1ull);\n"
         "  const u64 idx = atomicAdd(qcount, 1ull);\n"
         "  if (idx < qcap) { qent e; e.key = (q << 48) | pos; e.bits = bits; e.pad = 0u; queue[idx] = e; }\n"
         "}\n";
    // Hand-over of one hit record to the host's pinned buffer: stores of different waves (XCDs) reach host memory in
    // no particular order relative to the last wave's sequence word, and the two 16-byte halves of a record are
    // separate stores -- so EACH half carries the scan's tag and is written by ONE dwordx4 store:
    //   half 0 = {pos | (seq & 0xFFFFFF) << 40, record, pattern}     (positions are < 2^40: 288 GB of HBM)
    //   half 1 = {m0, slot | seq << 32}                               (m1 is zero for patterns <= 64 nt)
    // The host takes a record only when both tags are this scan's (host.cpp: scan_collect) and strips them.
    // withhold (tests only): slot + 1 of a record whose first half is published with a stale tag -- a torn record.
    s << "__device__ __forceinline__ void publish(hitrec* __restrict__ pub_hits, u64 slot, u64 pos, u32 record, u32 pattern,\n"
         "    u64 m0, u32 seq, u32 withhold) {\n"
         "  const u32 tag0 = (withhold != 0u && slot + 1ull == (u64)withhold) ? (seq - 1u) : seq;\n"
         "  const u64 tpos = pos | ((u64)(tag0 & 0xFFFFFFu) << 40);\n"
         "  v4 a, b;\n"
         "  a.x = (u32)tpos; a.y = (u32)(tpos >> 32); a.z = record; a.w = pattern;\n"
         "  b.x = (u32)m0; b.y = (u32)(m0 >> 32); b.z = (u32)slot; b.w = seq;\n"
         "  v4* dst = (v4*)(pub_hits + slot);\n"
         "  dst[0] = a; // one global_store_dwordx4 each (checked in the ISA by csrc/jit_check.py)\n"
         "  dst[1] = b;\n"
         "}\n";
    // the patterns of this kernel as constant tables (the verifier's per-position masks: low 4 bits =
    // bases the primer position accepts, bit 4 = position inside the protected window)
    s << "#define NPAT " << pats.size() << "u\n#define QBASE " << qbase << "u\n";
    s << "__device__ const unsigned char __attribute__((aligned(16))) PMASK[NPAT * 32u] = {";
    for (size_t q = 0; q < full_pats.size(); ++q)
        for (int j = 0; j < 32; ++j) s << (q || j ? "," : "") << (j < full_pats[q].len ? (unsigned)full_pats[q].mask[j] : 0u);
    s << "};\n__device__ const u32 PINFO[NPAT * 4u] = {"; // len, seed_off, seed_len, global_id
    for (size_t q = 0; q < full_pats.size(); ++q)
        s << (q ? "," : "") << full_pats[q].len << "u," << full_pats[q].seed_off << "u," << full_pats[q].seed_len << "u," << full_pats[q].global_id << "u";
    s << "};\n";
    s << "#define CAND_CAP " << CAND_CAP << "u\n";
    s << "// IPCR_WAVES_PER_GROUP " << WPG << "\n";
    s << "extern \"C\" __global__ void __launch_bounds__(" << WPG * 64 << ", " << WPS << ") ipcr_filter(const v4* __restrict__ planes, u64 block0, u64 nblocks,\n"
         "    qent* __restrict__ queue_all, u64 qcap, u64* __restrict__ qcount_all,\n"
         "    const u32* __restrict__ rst, const dpat* __restrict__ pats, const u64* __restrict__ rec_start,\n"
         "    const u64* __restrict__ rec_len, const u32* __restrict__ block_rec, u32 nrec, u32 max_mm, u32 check_rst,\n"
         "    hitrec* __restrict__ hits, u64 hcap,\n"
         "    u64* __restrict__ counts, u64* __restrict__ next_counts, u64* __restrict__ next_qcount,\n"
         "    u32* __restrict__ tickets, u64* __restrict__ pub, hitrec* __restrict__ pub_hits, u32 pre, u32* __restrict__ pub_seq, u32 seq,\n"
         "    u32 withhold) {\n";
    s << "  const u32 lane = threadIdx.x & 63u;\n";
    s << "  const u32 wv = threadIdx.x >> 6;\n";
    s << "  const u64 bidx = (u64)blockIdx.x * " << WPG << "u + wv; // this launch sweeps blocks [block0, block0 + nblocks)\n";
    s << "  const u64 nwv = nblocks * " << SEG << "ull; // waves that work in this launch\n";
    if (seg) {
        s << "  const u64 block = block0 + bidx / " << SEG << "ull;\n";
        s << "  const u32 sg = (u32)(bidx % " << SEG << "ull); // this wave's share of the block: windows that end in iterations [e0, e1)\n";
        s << "  const u32 seg_e0 = sg * " << NIT << "u / " << SEG << "u, seg_e1 = (sg + 1u) * " << NIT << "u / " << SEG << "u, seg_f0 = seg_e0 ? seg_e0 - 1u : 0u;\n";
    } else
        s << "  const u64 block = block0 + bidx;\n";
    // the counters alternate between two sets; workgroup 0 clears the set the NEXT scan will use
    s << "  if (blockIdx.x == 0u && next_counts) {\n"
         "    if (threadIdx.x < 4u) next_counts[threadIdx.x] = 0ull;\n"
         "    for (u32 t = threadIdx.x; t < 256u; t += " << WPG * 64 << "u) next_qcount[t * 16u] = 0ull;\n"
         "  }\n";
    s << "  if (bidx >= nwv) return;\n";
    s << "  __shared__ u64 lkey_all[" << WPG << "][LIST_CAP + 1u];\n";
    s << "  __shared__ u32 lbits_all[" << WPG << "][LIST_CAP + 1u];\n";
    s << "  __shared__ u32 lcnt_all[" << WPG << "];\n";
    s << "  __shared__ u64 candP_all[" << WPG << "][CAND_CAP];\n";
    s << "  __shared__ unsigned char candq_all[" << WPG << "][CAND_CAP];\n";
    s << "  u64* lkey = lkey_all[wv]; u32* lbits = lbits_all[wv]; u32* lcnt = lcnt_all + wv;\n";
    s << "  if (lane == 0u) *lcnt = 0u;\n";
    // this wave's copy of the pattern tables (tail verify: no global round trip in front of the tile loads)
    s << "  __shared__ u32 ptab_all[" << WPG << "][NPAT * 12u];\n";
    s << "  u32* ptab = ptab_all[wv];\n";
    s << "  for (u32 t = lane; t < NPAT * 12u; t += 64u) ptab[t] = t < NPAT * 8u ? ((const u32*)PMASK)[t] : PINFO[t - NPAT * 8u];\n";
    s << "  const u32 shard = (u32)block & 255u; // candidate queue: 256 segments, one push counter each\n";
    s << "  qent* queue = queue_all + (u64)shard * qcap;\n";
    s << "  u64* qcount = qcount_all + shard * 16u;\n";
    s << "  const v4* own = planes + block * 6144ull + lane;\n";
    s << "  const v4* nblk = planes + (block + 1ull) * 6144ull; // column 0 of the next block\n";
    s << "  const u64 posbase = ((block * 64ull + lane) * 32ull) << 7;\n";
    s << "  u32 ";
    for (int i = 0; i < W; ++i) {
        if (i) s << ", ";
        s << "a" << i << " = 0, c" << i << " = 0, g" << i << " = 0, t" << i << " = 0, n" << i << " = 0";
    }
    s << ";\n";
    const int SC = nb_lds ? 65 : 64; // stash columns: the wave's 64 + column 0 of the next block
    s << "  __shared__ v4 stash[" << WPG << "][" << QW * 3 << "][" << SC << "]; // head quads of each wave's block, for the wrap rows\n";
    s << "  v4 (*st)[" << SC << "] = stash[wv];\n";
    for (int i = 1; i <= D; ++i) {
        const std::string at = seg ? "(seg_f0 * " + std::to_string(QPI) + "u + " + std::to_string(i - 1) + "u) * 192u" : std::to_string((i - 1) * 192);
        s << "  v4 p" << i << "lo = own[" << at << "], p" << i << "hi = own[" << at << " + 64u], p" << i << "iv = own[" << at << " + 128u];\n";
    }
    if (nb_lds)
        s << "  v4 nbv = {0u, 0u, 0u, 0u}; // word (lane / 3, lane % 3) of the next block's column 0\n"
             "  if (lane < " << QW * 3 << "u) nbv = nblk[(lane / 3u) * 192u + (lane % 3u) * 64u];\n";
    if (seg) { // the head quads for the wrap rows: loaded by every wave whose walk reaches them (they are in the L2: the block's first wave streams them)
        s << "  if (seg_e1 * " << QPI << "u + " << D << "u > 32u) {\n";
        for (int q = 0; q < QW; ++q)
            s << "    st[" << q * 3 << "][lane] = own[" << q * 192 << "]; st[" << q * 3 + 1 << "][lane] = own[" << q * 192 + 64 << "]; st["
              << q * 3 + 2 << "][lane] = own[" << q * 192 + 128 << "];\n";
        if (nb_lds) s << "    if (lane < " << QW * 3 << "u) st[lane][64] = nbv;\n";
        s << "  }\n";
    }
    s << pro.str();
    if (nb_lds && peel) s << "  if (lane < " << QW * 3 << "u) st[lane][64] = nbv; // (loaded before iteration 0: it has long arrived)\n";
    if (seg) s << "  for (u32 it = seg_f0; it < seg_e1; ++it) {\n";
    else s << "  for (u32 it = " << (peel ? 1 : 0) << "u; it < " << NIT << "u; ++it) {\n";
    if (nb_lds && !peel && !seg) s << "    if (it == 1u && lane < " << QW * 3 << "u) st[lane][64] = nbv;\n";
    s << body.str() << "  }\n";
    s << epi.str();
    // ---- exact verification of this wave's survivors (verifyAt, core/engine/ac.go:186-213 / the
    // inner loop of FindMatches, core/primer/match.go:67-84): lane j compares window position j
    // Fast path (a few candidates per wave, the normal case): the surviving words are expanded into a
    // list of candidate windows, and every load round checks TWO of them, one per half-wave (patterns
    // are <= 32 nt); the record of the block is looked up once per wave.  Per candidate that is one
    // L2 round trip instead of the chain list -> pattern -> record search -> tiles.
    // Slow path (dense survivors: more than CAND_CAP windows): one window at a time, as before.
    s << "  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, \"wavefront\");\n"
         "  u32 nl = (u32)__builtin_amdgcn_readfirstlane(*lcnt);\n"
         "  if (nl > LIST_CAP) nl = LIST_CAP;\n"
         "  const u32* planes32 = (const u32*)planes;\n"
         "  u32 ncand = 0u;\n"
         "  bool wrote = false;\n"
         "  if (nl) {\n"
         "    u64* candP = candP_all[wv]; unsigned char* candq = candq_all[wv];\n"
         "    u64 ekey = 0ull; u32 ebits = 0u;\n"
         "    if (lane < nl) { ekey = lkey[lane]; ebits = lbits[lane]; }\n"
         "    const u32 mine = (u32)__builtin_popcount(ebits);\n"
         "    u32 incl = mine;\n"
         "    for (int d = 1; d < 64; d <<= 1) { const u32 up = __shfl_up(incl, d); if ((int)lane >= d) incl += up; }\n"
         "    ncand = (u32)__builtin_amdgcn_readlane((int)incl, 63);\n"
         "    if (ncand <= CAND_CAP) {\n"
         "      u32 o = incl - mine;\n"
         "      while (ebits) {\n"
         "        const u32 bit = (u32)__builtin_ctz(ebits); ebits &= ebits - 1u;\n"
         "        candP[o] = (ekey & 0xFFFFFFFFFFFFull) + ((u64)bit << 7); candq[o] = (unsigned char)((u32)(ekey >> 48) - QBASE); ++o;\n"
         "      }\n"
         "      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, \"wavefront\");\n"
         "      const u32 r0 = block_rec[block]; // record of the block's first base; candidates come in no particular order\n"
         "      const u64 rs0 = rec_start[r0], rl0 = rec_len[r0], rn0 = (r0 + 1u < nrec) ? rec_start[r0 + 1u] : ~0ull;\n"
         "      const u32 half = lane >> 5, j = lane & 31u;\n"
         "      for (u32 i = 0u; i < ncand; i += 2u) {\n"
         "        const u32 c = i + half;\n"
         "        const bool live = c < ncand;\n"
         "        const u64 P = live ? candP[c] : candP[i];\n"
         "        const u32 q = live ? candq[c] : candq[i];\n"
         "        const u32 g = base_bits(planes32, P + j); // always in bounds: >= 128 pad bases follow every record\n"
         "        const u32* pi = ptab + NPAT * 8u + q * 4u;\n"
         "        const u32 L = pi[0], soff = pi[1], slen = pi[2], gid = pi[3];\n"
         "        const u32 m = (ptab[q * 8u + (j >> 2)] >> ((j & 3u) * 8u)) & 0xFFu;\n"
         "        const u32 onehot = (g & 4u) ? 0u : (1u << (g & 3u));\n"
         "        const bool mis = j < L && (m & onehot) == 0u;\n"
         "        const bool prot = mis && (m & 16u);\n"
         "        u32 r = r0; u64 rs = rs0, rl = rl0, rn = rn0;\n"
         "        while (P >= rn) { ++r; rs = rn; rl = rec_len[r]; rn = (r + 1u < nrec) ? rec_start[r + 1u] : ~0ull; } // a block holds few records\n"
         "        const u64 local = P - rs;\n"
         "        const u32 mm = (u32)(__ballot(mis) >> (half * 32u));\n"
         "        const u32 pv = (u32)(__ballot(prot) >> (half * 32u));\n"
         "        const bool ok = live && pv == 0u && (u32)__builtin_popcount(mm) <= max_mm && P >= rs && local + L <= rl; // P < rs: a shifted window start in the padding in front of the record\n"
         "        u32 flag = 0u;\n"
         "        if (check_rst) flag = ((u32)(__ballot(ok && j < slen && rst_bit(rst, P + soff + j)) >> (half * 32u))) != 0u ? 1u : 0u;\n"
         "        if (__ballot(ok) != 0ull) wrote = true;\n"
         "        if (ok && j == 0u) {\n"
         "          const u64 slot = atomicAdd(counts + 1, 1ull);\n"
         "          hitrec h; h.pos = local; h.record = r; h.pattern = gid | (flag << 31); h.m0 = (u64)mm; h.m1 = 0ull;\n"
         "          if (slot < hcap) hits[slot] = h;\n"
         "          if (pub_hits && slot < pre) publish(pub_hits, slot, h.pos, h.record, h.pattern, h.m0, seq, withhold); // the first hits also go straight to the host\n"
         "        }\n"
         "      }\n"
         "      nl = 0u; // done\n"
         "    } else ncand = 0u;\n"
         "  }\n"
         "  for (u32 e = 0u; e < nl; ++e) {\n"
         "    const u64 key = lkey[e];\n"
         "    u32 bits = (u32)__builtin_amdgcn_readfirstlane(lbits[e]);\n"
         "    const u32 q = (u32)__builtin_amdgcn_readfirstlane((u32)(key >> 48));\n"
         "    const u64 P0 = ((u64)(u32)__builtin_amdgcn_readfirstlane((u32)(key >> 32) & 0xFFFFu) << 32) | (u64)(u32)__builtin_amdgcn_readfirstlane((u32)key); // the builtin returns int\n"
         "    ncand += (u32)__builtin_popcount(bits);\n"
         "    const dpat* pp = pats + q;\n"
         "    const u32 L = pp->len, slen = pp->seed_len, soff = pp->seed_off, gid = pp->global_id;\n"
         "    const u32 m = lane < L ? pp->mask[lane] : 0u;\n"
         "    while (bits) {\n"
         "      const u32 bit = (u32)__builtin_ctz(bits);\n"
         "      bits &= bits - 1u;\n"
         "      const u64 P = P0 + ((u64)bit << 7);\n"
         "      u32 lo = 0u, hi = nrec; // last record with start <= P\n"
         "      while (hi - lo > 1u) { const u32 mid = (lo + hi) >> 1; if (rec_start[mid] <= P) lo = mid; else hi = mid; }\n"
         "      const u64 local = P - rec_start[lo];\n"
         "      if (P < rec_start[lo] || local + L > rec_len[lo]) continue; // window must stay inside the record (ac.go:188-190)\n"
         "      bool mis = false, prot = false;\n"
         "      if (lane < L) {\n"
         "        const u32 g = base_bits(planes32, P + lane);\n"
         "        const u32 onehot = (g & 4u) ? 0u : (1u << (g & 3u));\n"
         "        mis = (m & onehot) == 0u;\n"
         "        prot = mis && (m & 16u);\n"
         "      }\n"
         "      const u64 mm = __ballot(mis);\n"
         "      if (__ballot(prot) != 0ull || (u32)__popcll(mm) > max_mm) continue;\n"
         "      u32 flag = 0u;\n"
         "      if (check_rst && slen) flag = __ballot(lane < slen && rst_bit(rst, P + soff + lane)) != 0ull ? 1u : 0u;\n"
         "      wrote = true;\n"
         "      if (lane == 0u) {\n"
         "        const u64 slot = atomicAdd(counts + 1, 1ull);\n"
         "        hitrec h; h.pos = local; h.record = lo; h.pattern = gid | (flag << 31); h.m0 = mm; h.m1 = 0ull;\n"
         "        if (slot < hcap) hits[slot] = h;\n"
         "        if (pub_hits && slot < pre) publish(pub_hits, slot, h.pos, h.record, h.pattern, h.m0, seq, withhold);\n"
         "      }\n"
         "    }\n"
         "  }\n"
         "  // candidate statistics: one same-address atomic per wave (~12 ns each, serialised in L2) would cost more\n"
         "  // than the verification itself once most waves have a candidate: 64 counters next to the tickets instead\n"
         "  if (lane == 0u && ncand) { if (tickets) atomicAdd(tickets + ((u32)block & 63u) * 32u + 1u, ncand); else atomicAdd(counts + 2, (u64)ncand); }\n";
    // ---- the last wave of the scan hands the counters to the host (no copy operation behind the
    // kernel): two-level ticket, 64 first-level counters 128 B apart so that the ~12 ns same-address
    // atomics of thousands of finishing waves do not queue up behind one another
    s << "  if (pub) {\n"
         "    // my counter atomics and my records in the host's (uncached) buffer must have been performed before\n"
         "    // my ticket; nothing of mine sits dirty in this XCD's L2 that the last wave reads, so no L2 write-back\n"
         "    // (thousands of waves issuing one cost half the kernel time again)\n"
         "    if (wrote || ncand) asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n"
         "    const u32 sh = (u32)bidx & 63u;\n"
         "    const u32 expect = (u32)((nwv - sh + 63ull) >> 6); // waves of this ticket shard\n"
         "    u32 t = 0u;\n"
         "    if (lane == 0u) t = atomicAdd(tickets + sh * 32u, 1u);\n"
         "    t = (u32)__builtin_amdgcn_readfirstlane(t);\n"
         "    if (t + 1u == expect) {\n"
         "      u32 t2 = 0u;\n"
         "      if (lane == 0u) { tickets[sh * 32u] = 0u; t2 = atomicAdd(tickets + 2048u, 1u); }\n"
         "      t2 = (u32)__builtin_amdgcn_readfirstlane(t2);\n"
         "      const u32 nsh = nwv < 64ull ? (u32)nwv : 64u;\n"
         "      if (t2 + 1u == nsh) { // every other wave of the scan has finished\n"
         "        __threadfence();\n"
         "        u32 cs = __hip_atomic_load(tickets + lane * 32u + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // candidate windows, 64 partial counts\n"
         "        tickets[lane * 32u + 1u] = 0u;\n"
         "        for (int off = 32; off > 0; off >>= 1) cs += __shfl_down(cs, off);\n"
         "        const u32 cs_total = (u32)__builtin_amdgcn_readfirstlane((int)cs);\n"
         "        if (lane < 4u) {\n"
         "          u64 v = __hip_atomic_load(counts + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);\n"
         "          if (lane == 2u) { v += (u64)cs_total; counts[2] = v; }\n"
         "          pub[lane] = v;\n"
         "        }\n"
         "        if (lane == 0u) tickets[2048u] = 0u;\n"
         "        __threadfence_system();\n"
         "        if (lane == 0u) __hip_atomic_store(pub_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);\n"
         "      }\n"
         "    }\n"
         "  }\n";
    s << "}\n";
    (void)uses_n;
    return s.str();
}

namespace {

// code objects of this process, keyed by (arch, source): scanning several genomes, or the same
// primers under another HitCap / length window, does not pay hiprtc twice
struct CodeCache {
    std::mutex mu;
    std::map<std::string, std::vector<char>> map;
    size_t bytes = 0;
};
CodeCache &code_cache() {
    static CodeCache c;
    return c;
}

bool compile_group_uncached(const std::string &src, const std::string &arch, std::vector<char> &code, std::string &err);

// Where code objects persist between processes: a cold `ipcr` run of a panel that has been scanned before skips hiprtc
// (C2: 1.9 s) and pays the module load only.  IPCR_JIT_CACHE_DIR=<dir>, or "" to turn the disk cache off; default
// $XDG_CACHE_HOME/ipcr_hip, else $HOME/.cache/ipcr_hip (created on first use; a directory that cannot be written
// simply caches nothing).
std::string jit_cache_dir() {
    if (const char *dir = getenv("IPCR_JIT_CACHE_DIR")) return dir;
    std::string base;
    if (const char *x = getenv("XDG_CACHE_HOME")) base = x;
    if (base.empty())
        if (const char *h = getenv("HOME")) if (*h) base = std::string(h) + "/.cache";
    if (base.empty()) return "";
    const std::string dir = base + "/ipcr_hip";
    static std::once_flag once;
    std::call_once(once, [&] { (void)mkdir(base.c_str(), 0700); (void)mkdir(dir.c_str(), 0700); });
    return dir;
}

// The directory is bounded: after a new file has been written, the least recently used code objects (mtime: a hit touches
// its file) are removed until the ipcr_*.jit files take at most IPCR_JIT_CACHE_MAX_MB (default 256; a C2-sized kernel is
// ~0.2 MB, a 1024-row index kernel ~0.1 MB) -- a library must not grow a user's home directory without limit.
void prune_cache_dir(const std::string &dir) {
    if (dir.empty()) return;
    const unsigned long long limit = (unsigned long long)env_int("IPCR_JIT_CACHE_MAX_MB", 256, 1, 1 << 20) << 20;
    DIR *d = opendir(dir.c_str());
    if (!d) return;
    struct Ent { std::string path; unsigned long long size; long long mtime_ns; };
    std::vector<Ent> ents;
    unsigned long long total = 0;
    while (const dirent *de = readdir(d)) {
        const std::string nm = de->d_name;
        if (nm.size() < 10 || nm.compare(0, 5, "ipcr_") != 0 || nm.compare(nm.size() - 4, 4, ".jit") != 0) continue;
        struct stat sb;
        const std::string path = dir + "/" + nm;
        if (stat(path.c_str(), &sb) != 0 || !S_ISREG(sb.st_mode)) continue;
        ents.push_back({path, (unsigned long long)sb.st_size, (long long)sb.st_mtim.tv_sec * 1000000000ll + sb.st_mtim.tv_nsec});
        total += (unsigned long long)sb.st_size;
    }
    closedir(d);
    if (total <= limit) return;
    std::sort(ents.begin(), ents.end(), [](const Ent &a, const Ent &b) { return a.mtime_ns < b.mtime_ns; });
    for (const Ent &e : ents) {
        if (total <= limit) break;
        if (remove(e.path.c_str()) == 0) total -= e.size; // (another process may have removed it already: nothing lost)
    }
}

// hiprtc -> code object for one group (no device needed except for the arch name)
bool compile_group(const std::string &src, const std::string &arch, std::vector<char> &code, std::string &err) {
    const std::string key = arch + "\n" + src;
    CodeCache &cc = code_cache();
    const bool memcache = !(getenv("IPCR_JIT_NO_MEMCACHE") && atoi(getenv("IPCR_JIT_NO_MEMCACHE"))); // (off: bench.py measures what a fresh process pays)
    if (memcache) {
        std::lock_guard<std::mutex> lk(cc.mu);
        const auto it = cc.map.find(key);
        if (it != cc.map.end()) { code = it->second; return true; }
    }
    // IPCR_JIT_CACHE_DIR: code objects also persist on disk (a CLI that is run again with the same primers skips
    // hiprtc).  File name = 64-bit FNV-1a of the key; the file STORES its key (arch, hiprtc version, source) in front
    // of the code object and is used only when that key is byte-identical: a hash collision or a code object built
    // by another ROCm release is compiled afresh, never loaded silently.
    std::string disk, fullkey;
    {
        const std::string cache_dir = jit_cache_dir();
        const char *dir = cache_dir.c_str();
        if (*dir) {
            // the whole version: hiprtc's major.minor, the HIP runtime's number (patch level included) and the headers
            // this library was built against -- a code object of another ROCm build is compiled afresh, never loaded
            int vmaj = 0, vmin = 0, vrt = 0;
            (void)hiprtcVersion(&vmaj, &vmin);
            (void)hipRuntimeGetVersion(&vrt);
            fullkey = "hiprtc " + std::to_string(vmaj) + "." + std::to_string(vmin) + " runtime " + std::to_string(vrt) + " built " +
                      std::to_string(HIP_VERSION) + "\n" + key;
            unsigned long long h = 1469598103934665603ull;
            for (const unsigned char ch : fullkey) { h ^= ch; h *= 1099511628211ull; }
            char name[64];
            snprintf(name, sizeof name, "/ipcr_%016llx.jit", h);
            disk = std::string(dir) + name;
            if (FILE *fh = fopen(disk.c_str(), "rb")) {
                std::vector<char> buf;
                char tmp[1 << 16];
                size_t n;
                while ((n = fread(tmp, 1, sizeof tmp, fh)) > 0) buf.insert(buf.end(), tmp, tmp + n);
                fclose(fh);
                // layout: "IPCRJIT1" | u64 key length | key | code object
                unsigned long long klen = 0;
                if (buf.size() > 16 && memcmp(buf.data(), "IPCRJIT1", 8) == 0) memcpy(&klen, buf.data() + 8, 8);
                if (klen == fullkey.size() && buf.size() > 16 + klen + 64 && memcmp(buf.data() + 16, fullkey.data(), klen) == 0 &&
                    memcmp(buf.data() + 16 + klen, "\177ELF", 4) == 0) {
                    code.assign(buf.begin() + 16 + (long)klen, buf.end());
                    (void)utimensat(AT_FDCWD, disk.c_str(), nullptr, 0); // used now: the pruning below removes the least recently used first
                }
            }
        }
    }
    if (code.empty()) {
        if (!compile_group_uncached(src, arch, code, err)) return false;
        if (!disk.empty()) { // write next to the target, then rename: readers never see a partial file
            const std::string tmpname = disk + ".tmp" + std::to_string((unsigned long long)getpid());
            if (FILE *fh = fopen(tmpname.c_str(), "wb")) {
                const unsigned long long klen = fullkey.size();
                bool ok = fwrite("IPCRJIT1", 1, 8, fh) == 8 && fwrite(&klen, 1, 8, fh) == 8 &&
                          fwrite(fullkey.data(), 1, fullkey.size(), fh) == fullkey.size() &&
                          fwrite(code.data(), 1, code.size(), fh) == code.size();
                ok = (fclose(fh) == 0) && ok;
                if (!ok || rename(tmpname.c_str(), disk.c_str()) != 0) (void)remove(tmpname.c_str());
                else prune_cache_dir(jit_cache_dir());
            }
        }
    }
    std::lock_guard<std::mutex> lk(cc.mu);
    if (cc.bytes > ((size_t)256 << 20)) { cc.map.clear(); cc.bytes = 0; }
    cc.bytes += key.size() + code.size();
    cc.map.emplace(key, code);
    return true;
}

bool compile_group_uncached(const std::string &src, const std::string &arch, std::vector<char> &code, std::string &err) {
    hiprtcProgram prog = nullptr;
    if (hiprtcCreateProgram(&prog, src.c_str(), "ipcr_filter.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) {
        err = "hiprtcCreateProgram failed";
        return false;
    }
    const std::string archopt = "--offload-arch=" + arch;
    // -O2: the generated code is straight-line explicit instructions -- measured on the C2 / C3 sources: the same instruction
    // count and registers as -O3 (16 328 against 16 338, 203 VGPRs), 7-12 % less compile time; nothing to vectorise: 15 %
    const char *opts[] = {archopt.c_str(), "-O2", "-fno-slp-vectorize", "-fno-vectorize"};
    const hiprtcResult r = hiprtcCompileProgram(prog, 4, opts);
    if (r != HIPRTC_SUCCESS) {
        size_t n = 0;
        hiprtcGetProgramLogSize(prog, &n);
        std::string log(n, '\0');
        if (n) hiprtcGetProgramLog(prog, &log[0]);
        err = std::string("hiprtc: ") + hiprtcGetErrorString(r) + "\n" + log;
        hiprtcDestroyProgram(&prog);
        return false;
    }
    size_t csize = 0;
    hiprtcGetCodeSize(prog, &csize);
    code.resize(csize);
    hiprtcGetCode(prog, code.data());
    hiprtcDestroyProgram(&prog);
    return true;
}

} // namespace

size_t jit_group_size(const std::vector<ipcr_dev_pattern> &pats, int max_mm) {
    for (const auto &p : pats)
        if (p.len == 0 || p.len > 128) return 0; // not specialisable: table-driven filter
    if (pats.empty()) return 0;
    // the unrolled main loop of one kernel should stay inside the 64 KiB instruction cache:
    // ~200 B of hot code per pattern per row step, W row steps
    const size_t G = (size_t)env_int("IPCR_JIT_GROUP", 12, 1, 48);
    const size_t ngroups = (pats.size() + G - 1) / G;
    // every group is a separate hiprtc compile (seconds each) and a sweep of its own: beyond 8 groups the
    // seed-index kernel serves (one compile, one sweep of ~8 ms per 3 Gb whatever the panel) -- but it takes panels
    // of k <= 3 only (host.cpp: panel_upload).  With more mismatches the alternative is the table-driven filter,
    // ~1.6 ms per pattern and 3 Gb: there 32 groups (384 patterns; ~10 s of compiles, 32 sweeps of ~0.25 ms) are the
    // better trade
    if (ngroups > (size_t)env_int("IPCR_JIT_MAX_GROUPS", max_mm > 3 ? 32 : 8, 1, 4096)) return 0;
    return (pats.size() + ngroups - 1) / ngroups; // balanced
}

// Large panels are cut into groups of patterns; every group becomes its own kernel (each
// streams the tiles once and appends to the same candidate queue).  Groups compile in parallel.
std::vector<JitFilter *> jit_build(const std::vector<ipcr_dev_pattern> &all_pats, int max_mm, std::string &err,
                                   const std::vector<uint32_t> *subset) {
    std::vector<JitFilter *> out;
    std::vector<ipcr_dev_pattern> chosen;
    if (subset)
        for (uint32_t q : *subset) chosen.push_back(all_pats[q]);
    const std::vector<ipcr_dev_pattern> &pats = subset ? chosen : all_pats;
    const size_t G = jit_group_size(pats, max_mm);
    if (G == 0) { err = "panel not specialised (more pattern groups than IPCR_JIT_MAX_GROUPS)"; return out; }
    const size_t ngroups = (pats.size() + G - 1) / G;
    int dev = 0;
    (void)hipGetDevice(&dev);
    hipDeviceProp_t prop;
    std::string arch = "gfx950";
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.gcnArchName[0]) {
        arch = prop.gcnArchName;
        const size_t colon = arch.find(':');
        if (colon != std::string::npos) arch = arch.substr(0, colon);
    }
    std::vector<std::string> srcs(ngroups), errs(ngroups);
    std::vector<std::vector<char>> codes(ngroups);
    std::vector<char> ok(ngroups, 0);
    for (size_t g = 0; g < ngroups; ++g) {
        const size_t q0 = g * G, q1 = std::min(pats.size(), q0 + G);
        if (subset) {
            const std::vector<uint32_t> ids(subset->begin() + (long)q0, subset->begin() + (long)q1);
            srcs[g] = jit_source(std::vector<ipcr_dev_pattern>(pats.begin() + (long)q0, pats.begin() + (long)q1), max_mm, 0, &ids, true);
        } else
            srcs[g] = jit_source(std::vector<ipcr_dev_pattern>(pats.begin() + (long)q0, pats.begin() + (long)q1), max_mm, (unsigned)q0);
    }
    unsigned nthreads = std::thread::hardware_concurrency();
    if (nthreads == 0) nthreads = 4;
    nthreads = (unsigned)std::min<size_t>({(size_t)nthreads, (size_t)16, ngroups});
    std::atomic<size_t> next{0};
    auto worker = [&]() {
        for (;;) {
            const size_t g = next.fetch_add(1);
            if (g >= ngroups) break;
            ok[g] = compile_group(srcs[g], arch, codes[g], errs[g]) ? 1 : 0;
        }
    };
    if (nthreads <= 1) worker();
    else {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nthreads; ++t) th.emplace_back(worker);
        for (auto &t : th) t.join();
    }
    for (size_t g = 0; g < ngroups; ++g)
        if (!ok[g]) { err = errs[g]; return out; }
    for (size_t g = 0; g < ngroups; ++g) {
        JitFilter *f = new JitFilter;
        {
            const size_t q0 = g * G, q1 = std::min(pats.size(), q0 + G);
            f->pats.assign(pats.begin() + (long)q0, pats.begin() + (long)q1);
            f->has_ids = subset != nullptr;
            if (subset) f->ids.assign(subset->begin() + (long)q0, subset->begin() + (long)q1);
            f->spill_only = subset != nullptr;
            f->max_mm = max_mm;
            f->qbase = subset ? 0u : (unsigned)q0;
            f->arch = arch;
        }
        const size_t at = srcs[g].find("// IPCR_WAVES_PER_GROUP ");
        if (at != std::string::npos) f->waves_per_group = (unsigned)atoi(srcs[g].c_str() + at + 24);
        if (hipModuleLoadData(&f->module, codes[g].data()) != hipSuccess ||
            hipModuleGetFunction(&f->fn, f->module, "ipcr_filter") != hipSuccess) {
            err = "hipModuleLoadData/GetFunction failed for the specialised filter";
            jit_destroy(f);
            for (JitFilter *o : out) jit_destroy(o);
            out.clear();
            return out;
        }
        out.push_back(f);
    }
    return out;
}

// ---- seed-index filter with the panel's key shapes baked in (device_types.h: ipcr_index_shape).
// One thread per strand walks its 128 rows plus `tail_rows` rows of the next strand with a rolling 2-bit k-mer.
// Per base step every shape is looked up in its LDS bitmap: the word index is the shape's block field of the k-mer,
// the bit the protected bases (FAST groups extract those six bits once for all their shapes and read the 32-bit
// half word that holds the bit; other groups compute the whole key per shape).  One ds_read_b32 per shape and step is
// all the LDS traffic of a step without a hit -- the kernel is bound by LDS cycles and VALU issue together, no
// registers are spent on delaying words, and the step knows WHICH shapes hit.
//  No validity test here: a window with <= k mismatches has a key whose bases are all valid and exact, so invalid
//  bases (code A in the k-mer) can only add candidates, which the exact check below rejects.
//  * hits go to a per-wave LDS queue (k-mer, invalid flags, where | shape bits) -- ballot + mbcnt slots, no atomics --
//    and are drained 64 at a time: each lane ranks its hit's key among the shape's keys (prefix per 64-bit word +
//    popcount), loads the entry and checks the pattern exactly against the k-mer.
static unsigned index_waves() { return (unsigned)env_int("IPCR_INDEX_WAVES", 16, 4, 16) / 4u * 4u; } // one workgroup per CU (the LDS image is staged once per CU), 4 waves per SIMD unless the dev knob says otherwise
#define IPCR_INDEX_WAVES index_waves()
static unsigned index_words64(const std::vector<ipcr_index_shape> &shapes) {
    unsigned t = 0;
    for (const ipcr_index_shape &x : shapes) t += ipcr_index_words64(x);
    return t;
}
static bool index_paired(const std::vector<ipcr_index_shape> &shapes) { return !shapes.empty() && shapes[0].paired != 0; } // all of a panel's shapes or none
static unsigned index_image_bytes(const std::vector<ipcr_index_shape> &shapes) { // bitmaps + rank prefixes (uint16 per 64 keys) + first entries + shape constants (build_index)
    const unsigned per64 = index_paired(shapes) ? 9u : 10u; // two-step tables: 64 keys are four 32-bit words, half of each a copy for the other step
    const unsigned image = index_words64(shapes) * per64 + (unsigned)shapes.size() * 16u + 32u; // (+ the drain's bit table)
    return (image + 15u) & ~15u;
}
unsigned jit_index_image_bytes(const std::vector<ipcr_index_shape> &shapes) { return index_image_bytes(shapes); }
// Two steps per lookup: steps per queue entry and entry layout for n shapes and windows that reach TR bases back (0: not possible)
static unsigned paired_plan(size_t ns, int TR, unsigned *mode_out) {
    const unsigned ROWB = 7;
    for (unsigned spe : {4u, 2u}) {
        if (ns * spe > 31u) continue; // payload bits 0..30, one per (shape, step)
        const unsigned nb = (unsigned)(TR < 0 ? 0 : TR) + spe;
        if (2u * nb + 6u + ROWB <= 64u) { if (mode_out) *mode_out = 0; return spe; }
        if (2u * nb + ROWB <= 64u && nb + 6u <= 32u) { if (mode_out) *mode_out = 1; return spe; }
    }
    return 0;
}
bool jit_index_pairable(size_t n_shapes, int tail_rows) {
    if (env_int("IPCR_INDEX_TWO_STEP", 0, 0, 1) == 0 || n_shapes == 0) return false; // off unless asked for: measured 18 % SLOWER (below)
    return paired_plan(n_shapes, tail_rows, nullptr) != 0;
}
static unsigned index_queue_entries(const std::vector<ipcr_index_shape> &shapes) { // per-wave hit queue: what the image leaves of the 160 KiB, in rounds of 64
    const unsigned left = 160u * 1024u - std::min(160u * 1024u, index_image_bytes(shapes));
    unsigned q = left / (IPCR_INDEX_WAVES * 16u) / 64u * 64u;
    return q < 128u ? 128u : (q > 448u ? 448u : q);
}

// ---- the k-mers are TRANSPOSED, not rolled.
// One unit of work = one column pair = 64 consecutive strands, lane = strand, 128 base steps, no tail rows: a lane starts
// with the 32 bases in front of its strand (the end of the strand before) in its k-mer, so every window is looked up
// exactly once, by the lane in whose strand it ENDS (right-anchored groups) / where its start + DL falls (left-anchored).
// A strand's bases are bit columns of the tiles; a rolled k-mer takes them out one row at a time (two v_bfe, a 64-bit
// shift, two shift-ors per step, and the same again for the invalid flags: 26 issue cycles of the ~130 a step costs).
// Here the 32 lanes of a half wave load the 32 rows of a chunk, one word each -- lane j the plane j & 1 of row
// 15 - (j >> 1) -- and a 32 x 32 bit transpose across the lanes (five butterfly stages: ds_swizzle xor d, a rotate and one
// v_bitop3 select) hands every lane the 16 bases of ITS strand as one register, 2-bit codes interleaved, newest base in
// bits 1:0; three transposes (two k-mer words, one word of invalid flags) per 32 steps.  Every key field of every step
// then sits at a compile-time position of three registers: a shift (or v_alignbit) and a mask.
std::string jit_index_source(const std::vector<ipcr_index_shape> &shapes, const IndexGeom &geom) {
    const int tail_rows = geom.tail_rows;
    const bool all_acgt = geom.all_acgt;
    const size_t NS = shapes.size();
    struct Grp { bool fast = false; std::vector<int> sh; int c_off = 0; };
    std::vector<Grp> groups;
    std::vector<unsigned> off64(NS + 1, 0); // first 64-bit word of every shape's bitmap
    for (size_t i = 0; i < NS; ++i) {
        if (shapes[i].group >= groups.size()) groups.resize((size_t)shapes[i].group + 1);
        groups[shapes[i].group].sh.push_back((int)i);
        groups[shapes[i].group].fast = shapes[i].fast != 0 && shapes[i].tw_bits >= 6;
        groups[shapes[i].group].c_off = shapes[i].tw_shift;
        off64[i + 1] = off64[i] + ipcr_index_words64(shapes[i]);
    }
    const unsigned T64N = off64[NS];
    const unsigned U = 32; // base steps per copy of the loop body = one chunk of rows
    const int TR = tail_rows < 0 ? 0 : tail_rows;
    // ---- how a base step reports its key hits.  Shapes are looked up in PACKS: the shapes of a FAST group share the
    // six key bits of the protected bases, so their bitmap BYTES (ds_read_u8: byte = key >> 3) are put side by side in
    // one register and tested with ONE shift by the shared low three key bits and one AND -- bits 0, 8, 16, 24 of the
    // result say which shapes hold the key; a shape outside a fast group is a pack of its own (one bit).  Pack p's bits
    // land on 8 i + p.  A lane files one queue entry per SPE base steps: the steps' masks are shifted together by SH =
    // number of packs, so bit 8 i + p + SH j = shape i of pack p, j steps before the entry's row.
    // Queue entry (16 bytes): x, y = the k-mer (newest base in bits 1:0); only the 2 NBAS bits a check can reach are kept,
    // z = the invalid-base flags of those bases, w = the steps' masks (bit 31 clear) or, for a further pattern chained
    // under a key, bit 31 | entry index << 2 | steps back.  Lane (6 bits) and row (ROWB bits) go where bits are left:
    //   layout A: both above the k-mer in y;  B: row above the k-mer in y, lane above the flags in z;
    //   C (primers beyond 26 nt): both in w above a 15-bit payload, one step per entry, one bit per shape.
    struct Pack { bool fast = false; int group = 0; std::vector<int> sh; };
    std::vector<Pack> packs;
    auto make_packs = [&](bool allow_fast) {
        packs.clear();
        for (size_t gi = 0; gi < groups.size(); ++gi) {
            const Grp &g = groups[gi];
            if (allow_fast && g.fast && g.sh.size() > 1) {
                for (size_t a = 0; a < g.sh.size(); a += 4) {
                    Pack p; p.fast = true; p.group = (int)gi;
                    for (size_t b = a; b < std::min(g.sh.size(), a + 4); ++b) p.sh.push_back(g.sh[b]);
                    packs.push_back(p);
                }
            } else
                for (int sidx : g.sh) { Pack p; p.fast = false; p.group = (int)gi; p.sh.push_back(sidx); packs.push_back(p); }
        }
    };
    make_packs(env_int("IPCR_INDEX_PACKED", 1, 0, 1) != 0);
    if (packs.size() > 8) make_packs(false);
    bool byte_layout = false;
    for (const Pack &p : packs) byte_layout |= p.sh.size() > 1;
    const unsigned ROWB = 7; // rows 0..127 of the unit
    unsigned SPE = 1, mode = 2;
    // ---- two steps per lookup (all shapes "3 protected bases + 5 block bases", host.cpp: build_index decides).  The keys of
    // steps t and t + 1 share two of their three protected bases and four of their five block bases: those twelve bits are
    // the ADDRESS of a 32-bit word; the base only step t has (the oldest of either field, four bits) picks one of its low 16
    // bits, the base only step t + 1 has (the newest of either field) one of its high 16 -- ONE ds_read_b32 per shape for
    // two base steps instead of two ds_read_u8; a shape's table is as large as a 17-bit bitmap: 16 KiB.
    // MEASURED (C4, 3 Gb): 6.09 ms per sweep against 5.15 with the same 16-bit keys looked up one step at a time (4.99 with
    // the default 17-bit keys) -- half the LDS lookups and the sweep is 18 % SLOWER.  What the pair saves in addresses
    // (3 per step instead of 6) it pays twice over in bit indices (one per shape and step, where a pack of byte lookups
    // shares ONE shift): +10 % VALU instructions, most of them four-cycle v_alignbit.  Splitting the bit collection over
    // 1 / 2 / 4 / 8 registers changes nothing (6.09 / 6.10 / 6.18 / 6.23): no latency chain -- the sweep follows the VALU
    // issue count, not the LDS instruction count.  Off by default (IPCR_INDEX_TWO_STEP=1), parity-tested.
    const bool paired = index_paired(shapes);
    if (paired) {
        SPE = paired_plan(NS, TR, &mode);
        if (SPE == 0) return std::string(); // (build_index asked jit_index_pairable first)
    } else {
        const unsigned SH0 = (unsigned)packs.size();
        unsigned want = byte_layout ? (SH0 <= 2 ? 4u : (SH0 <= 4 ? 2u : 1u)) : (SH0 * 4u <= 31u ? 4u : (SH0 * 2u <= 31u ? 2u : 1u));
        want = (unsigned)env_int("IPCR_INDEX_STEPS_PER_ENTRY", (int)want, 1, (int)want);
        while (want & (want - 1u)) --want;
        for (unsigned spe = want; spe >= 1; spe >>= 1) {
            const unsigned nb = (unsigned)TR + spe;
            unsigned top = 0; // highest payload bit: bit 31 is the chain flag
            for (unsigned p = 0; p < packs.size(); ++p) top = std::max(top, (byte_layout ? 8u * ((unsigned)packs[p].sh.size() - 1u) : 0u) + p + SH0 * (spe - 1u));
            if (top > 30u) continue;
            if (2u * nb + 6u + ROWB <= 64u) { SPE = spe; mode = 0; break; }
            if (2u * nb + ROWB <= 64u && nb + 6u <= 32u) { SPE = spe; mode = 1; break; }
        }
        if (mode == 2) { // long primers: the whole k-mer and all its flags are needed
            SPE = 1;
            if (byte_layout) { make_packs(false); byte_layout = false; }
        }
    }
    const unsigned NPK = (unsigned)packs.size(), SHF = NPK;
    const unsigned NBAS = (unsigned)TR + SPE;
    const unsigned KMHI = 2u * NBAS > 32u ? 2u * NBAS - 32u : 0u;
    auto bitpos = [&](unsigned p, unsigned i, unsigned j) { return (byte_layout ? 8u * i : 0u) + p + SHF * j; };
    std::vector<int> tab(32, 0);
    // the tests of a block shift their bits into NCH registers in turn (one chain of 24 dependent shifts is latency, not work)
    unsigned NCH = paired ? (unsigned)env_int("IPCR_INDEX_ACC_CHAINS", 1, 1, 8) : 1u;
    while (paired && ((unsigned)NS * SPE) % NCH) --NCH;
    const unsigned NPC = paired ? (unsigned)NS * SPE / NCH : 0u; // tests per chain
    const bool use_add = env_int("IPCR_INDEX_ACC_ADD", 0, 0, 1) != 0; // dev knob: a chain grows by add + v_bitop3 (from bit 0 up) instead of one v_alignbit (from bit 31 down)
    if (paired) { // the c-th (shape, step) test of an entry's block, in the order the code below makes them -> payload bit (c % NCH) * NPC + c / NCH
        unsigned c = 0;
        for (unsigned pi = 0; pi < SPE / 2u; ++pi)
            for (const Grp &g : groups)
                for (int si : g.sh)
                    for (unsigned o = 0; o < 2u; ++o, ++c) tab[(c % NCH) * NPC + (use_add ? NPC - 1u - c / NCH : c / NCH)] = si | (int)((SPE - 1u - (2u * pi + o)) << 4);
    } else
    for (unsigned p = 0; p < NPK; ++p)
        for (unsigned i = 0; i < packs[p].sh.size(); ++i)
            for (unsigned j = 0; j < SPE; ++j) tab[bitpos(p, i, j)] = packs[p].sh[i] | (int)(j << 4);
    const unsigned idx_bits = mode == 2 ? 13u : 29u;
    const bool chain_carry = geom.table_entries < (1u << idx_bits);
    const unsigned QCAP = index_queue_entries(shapes);

    std::ostringstream s;
    s << "// generated by ipcr_amd/csrc/jit.cpp: seed-index filter (transposed k-mers), " << NS << " key shapes in " << groups.size() << " groups / " << NPK
      << " packs, 128 rows per strand, " << SPE << " steps per queue entry (entry layout " << "ABC"[mode] << ")" << (paired ? ", two steps per lookup" : "") << "\n";
    s << "#ifndef __HIPCC_RTC__\n#include <hip/hip_runtime.h>\n#endif\n";
    s << "typedef unsigned int u32;\ntypedef unsigned long long u64;\ntypedef long long i64;\n";
    s << "typedef u32 v4 __attribute__((ext_vector_type(4)));\n";
    s << "struct qent { u64 key; u32 bits; u32 pad; };\n";
    s << "struct hitrec { u64 pos; u32 record; u32 pattern; u64 m0, m1; };\n";
    s << "struct dpat { unsigned short len, seed_off, seed_len, reserved; u32 global_id; unsigned char mask[128]; };\n";
    // FUSED (hits != null): the index's check is exact for every pattern it serves (<= 32 nt), so a window that passes IS a
    // match unless it crosses a record end -- the lane that found it looks the record up, turns the mismatch word into
    // positions, reads the seed span's reset bits where the window holds an invalid base at all, and appends the hit record
    // itself (device buffer + the tagged copy in the host's pinned buffer, as the specialised filter does): no global
    // candidate queue, no verify kernel, no copy operation behind the sweep.
    s << "struct fuse { const u32* rst; const dpat* pats; const u64* rec_start; const u64* rec_len; const u32* block_rec; u32 nrec, check_rst;\n"
         "  hitrec* hits; u64 hcap; u64* counts; hitrec* pub_hits; u32 pre, seq; };\n";
    s << "__device__ __forceinline__ u32 rst_bit(const u32* __restrict__ rst, u64 P) {\n"
         "  const u64 strand = P >> 7, col = strand >> 5;\n"
         "  const u32 row = (u32)P & 127u, bit = (u32)strand & 31u;\n"
         "  return (rst[((((col >> 6) * 32u + (row >> 2)) * 64u + (col & 63u)) << 2) + (row & 3u)] >> bit) & 1u;\n"
         "}\n"
         "// hand-over of one hit record to the host's pinned buffer: each 16-byte half carries the scan's tag and is ONE store\n"
         "// (jit_source: publish; host.cpp: scan_collect takes a record only when both tags are this scan's)\n"
         "__device__ __forceinline__ void publish(hitrec* __restrict__ pub_hits, u64 slot, u64 pos, u32 record, u32 pattern, u64 m0, u32 seq) {\n"
         "  const u64 tpos = pos | ((u64)(seq & 0xFFFFFFu) << 40);\n"
         "  v4 a, b;\n"
         "  a.x = (u32)tpos; a.y = (u32)(tpos >> 32); a.z = record; a.w = pattern;\n"
         "  b.x = (u32)m0; b.y = (u32)(m0 >> 32); b.z = (u32)slot; b.w = seq;\n"
         "  v4* dst = (v4*)(pub_hits + slot);\n"
         "  dst[0] = a;\n"
         "  dst[1] = b;\n"
         "}\n";
    s << "#define NS " << NS << "\n";
    s << "#define T64N " << T64N << "u // 64-bit words of all bitmaps\n";
    s << "#define QCAP " << QCAP << "u // per-wave queue of hits (16-byte entries), drained in rounds of 64 at full lane occupancy\n";
    s << "#define ALL_ACGT " << (all_acgt ? "true" : "false") << " // no indexed pattern holds an IUPAC code\n";
    s << "#define ULEN " << geom.uniform_len << "u // length of every indexed pattern (0: mixed)\n";
    bool aligned = geom.uniform_len > 0; // ... and left-anchored windows are tested exactly when they end: no shift in the check
    for (const ipcr_index_shape &x : shapes)
        if (x.left && (int)x.dl != geom.uniform_len - 1) aligned = false;
    s << "#define WINDOW_AT_NEWEST " << (aligned ? "true" : "false") << "\n";
    s << "#define DL " << geom.dl << "u // left-anchored windows are tested DL bases after their start\n";
    s << "#define CHAIN_CARRY " << (chain_carry ? "true" : "false") << " // further patterns of a key are handed back to the queue (entry index in " << idx_bits << " bits)\n";
    s << "#define SPE " << SPE << "u // base steps per queue entry\n";
    s << "#define ROWB " << ROWB << "u // bits of a row number\n";
    s << "#define KMHI " << KMHI << "u // k-mer bits an entry keeps in y\n";
    s << "#define NBAS " << NBAS << "u // bases (and invalid flags) an entry keeps\n";
    s << "#define EMODE " << mode << " // entry layout: 0 = lane, row above the k-mer in y; 1 = row in y, lane above the flags in z; 2 = both in w\n";
    s << "#define PAIRED " << (paired ? "true" : "false") << " // a shape's table: 2^12 words, low half = the keys of the older step of a pair, high half = of the newer\n";
    s << "#define PFX_WORDS " << (paired ? "(T64N / 4u)" : "(T64N / 2u)") << " // uint16 rank prefixes, one per 64 keys\n";
    s << "__device__ const unsigned char __attribute__((aligned(16))) BITTAB[32] = {";
    for (int b = 0; b < 32; ++b) s << (b ? ", " : "") << tab[(size_t)b];
    s << "}; // payload bit -> shape | steps back << 4\n";
    s << R"SRC(
// LDS image (host.cpp: build_index): the shapes' bitmaps, T64N 64-bit words | uint16 rank prefixes, one per 64 keys | NS x
// 4 words of shape constants (one ds_read_b128 in the drain: field shifts, field masks, first bitmap word, first entry);
// behind it (kernel start) the 32 bytes of BITTAB
#define PREFIX_WORD0 (T64N * 2u)
#define SHAPE_WORD0 (T64N * 2u + PFX_WORDS)
#define TAB_WORD0 (SHAPE_WORD0 + 4u * NS)
#define LDS_WORDS (TAB_WORD0 + 8u)
#define ANDOR(a, b, c) __builtin_amdgcn_bitop3_b32(a, b, c, 0xEA) /* (a & b) | c: v_bitop3_b32 issues in two cycles, v_and_or_b32 in four */
__device__ __forceinline__ u32 LSHL_OR(u32 a, u32 sh, u32 b) { u32 r; asm("v_lshl_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "n"(8), "v"(b)); (void)sh; return r; }
// invalid-base flags, one bit per base -> the even bits of a 2-bit-per-base word
__device__ __forceinline__ u64 spread2(u32 v) {
  u64 x = v;
  x = (x | (x << 16)) & 0x0000FFFF0000FFFFull;
  x = (x | (x << 8)) & 0x00FF00FF00FF00FFull;
  x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0Full;
  x = (x | (x << 2)) & 0x3333333333333333ull;
  x = (x | (x << 1)) & 0x5555555555555555ull;
  return x;
}
// 32 x 32 bit transpose across the 32 lanes of each half wave: lane i gives row i, lane s receives column s (bit i = bit s
// of row i).  Five butterfly stages: the partner's word (lane ^ d) comes through ds_swizzle, is rotated by d towards the
// half that is exchanged, and one v_bitop3 keeps the own bits under the lane's mask and takes the rest from the partner.
struct trc { u32 sh[5], mk[5]; };
template <int D> __device__ __forceinline__ u32 tr_stage(u32 x, u32 sh, u32 mk) {
  const u32 p = (u32)__builtin_amdgcn_ds_swizzle((int)x, (D << 10) | 0x1F);
  const u32 r = __builtin_amdgcn_alignbit(p, p, sh);
  return __builtin_amdgcn_bitop3_b32(x, r, mk, 0xE4); // (x & mk) | (r & ~mk)
}
__device__ __forceinline__ u32 tr32(u32 x, const trc& c) {
  x = tr_stage<16>(x, c.sh[0], c.mk[0]);
  x = tr_stage<8>(x, c.sh[1], c.mk[1]);
  x = tr_stage<4>(x, c.sh[2], c.mk[2]);
  x = tr_stage<2>(x, c.sh[3], c.mk[3]);
  x = tr_stage<1>(x, c.sh[4], c.mk[4]);
  return x;
}
// exact check of ONE pattern filed under a key (entry `idx`) against the k-mer of a hit; returns the entry of the
// next pattern with the same key (0xFFFFFFFF: none).
// Entry (device_types.h: ipcr_index_entry): {next, pattern, seq2 | prot2, len, flags | okA, okC | okG, okT}
__device__ __forceinline__ u32 check_entry(u32 idx, u64 km, u32 bad, int erow, u64 unit_base, u32 strand_off,
    u32 shard, const v4* __restrict__ table, u32 max_mm, qent* __restrict__ queue, u64 qcap, u64* __restrict__ qcount,
    const fuse& fu, u32& ncand) {
  const v4 e0 = table[idx * 4u], e1 = table[idx * 4u + 1u];
  const u64 seq2 = ((u64)e0.w << 32) | e0.z, prot2 = ((u64)e1.y << 32) | e1.x;
  const u32 left = e1.w & 1u;
  const u32 L = ULEN ? ULEN : e1.z;                 // one length for the whole panel: every shift and mask below is a constant
  const u32 sft = WINDOW_AT_NEWEST ? 0u : (left ? 2u * (DL + 1u - L) : 0u); // the window's last base sits sft / 2 bases behind the newest
  const u64 x = km >> sft;
  const u64 E = 0x5555555555555555ull;
  const u64 wmE = ((L >= 32u) ? ~0ull : ((1ull << (2u * L)) - 1ull)) & E;
  u64 mm2;
  if (ALL_ACGT || (e1.w & 2u)) {                     // one base per position: XOR against the primer's 2-bit codes
    const u64 d = x ^ seq2;
    mm2 = (d | (d >> 1)) & wmE;
  } else {                                           // IUPAC codes: four sets of allowed positions
    const v4 e2 = table[idx * 4u + 2u], e3 = table[idx * 4u + 3u];
    const u64 okA = ((u64)e2.y << 32) | e2.x, okC = ((u64)e2.w << 32) | e2.z;
    const u64 okG = ((u64)e3.y << 32) | e3.x, okT = ((u64)e3.w << 32) | e3.z;
    const u64 lo = x & E, hi = (x >> 1) & E;
    const u64 match = (~lo & ~hi & okA) | (lo & ~hi & okC) | (~lo & hi & okG) | (lo & hi & okT);
    mm2 = ~match & wmE;
  }
  const u32 bw = (bad >> (sft >> 1)) & ((L >= 32u) ? 0xFFFFFFFFu : ((1u << L) - 1u));
  if (bw) mm2 |= spread2(bw);                        // rare: the window holds an invalid base
  const int srow = left ? erow - (int)DL : erow - (int)L + 1; // may lie in the strand before (the padded coordinate is continuous)
  if ((mm2 & prot2) == 0ull && (u32)__popcll(mm2) <= max_mm) {
    // position = (the unit's first strand + the lane) * 128 + row
    const i64 Ps = (i64)unit_base + (i64)((int)strand_off + srow);
    if (fu.hits) { // the match is written here: record, bounds (core/engine/ac.go:188-190), mismatch positions, seed-span flag
      ++ncand;
      if (Ps >= 0) {
        const u64 P = (u64)Ps;
        u32 r = fu.block_rec[P >> 18]; // last record that starts at or before the block's first base
        u64 rs = fu.rec_start[r];
        while (r + 1u < fu.nrec) { const u64 rn = fu.rec_start[r + 1u]; if (P < rn) break; ++r; rs = rn; } // a block holds few records
        if (P >= rs && P - rs + L <= fu.rec_len[r]) { // (P < rs: a window that starts in the padding in front of the record)
          u64 m = 0ull; // mm2: position j at bit 2 (L - 1 - j)
          for (u64 t2 = mm2; t2 != 0ull; t2 &= t2 - 1ull) m |= 1ull << (L - 1u - ((u32)__builtin_ctzll(t2) >> 1));
          const dpat* pp = fu.pats + e0.y;
          u32 flag = 0u;
          if (fu.check_rst && bw) { // a reset byte is an invalid base: only a window that holds one can have one in its seed span
            const u32 soff = pp->seed_off, slen = pp->seed_len;
            for (u32 j = 0; j < slen; ++j) if (rst_bit(fu.rst, P + soff + j)) { flag = 1u; break; }
          }
          const u64 slot = atomicAdd(fu.counts + 1, 1ull);
          hitrec h; h.pos = P - rs; h.record = r; h.pattern = pp->global_id | (flag << 31); h.m0 = m; h.m1 = 0ull;
          if (slot < fu.hcap) fu.hits[slot] = h;
          if (fu.pub_hits && slot < (u64)fu.pre) publish(fu.pub_hits, slot, h.pos, h.record, h.pattern, h.m0, fu.seq);
        }
      }
    } else {
      const u64 qi = atomicAdd(qcount + shard * 16u, 1ull);
      if (qi < qcap) { qent qe; qe.key = ((u64)e0.y << 48) | (u64)Ps; qe.bits = 1u; qe.pad = 0u; queue[(u64)shard * qcap + qi] = qe; }
    }
  }
  return e0.x;
}
)SRC";
    s << "extern \"C\" __global__ void __launch_bounds__(" << IPCR_INDEX_WAVES * 64u << ", " << IPCR_INDEX_WAVES / 4u << ") ipcr_index_filter(const u32* __restrict__ planes, u64 cp0, u64 ncolpairs, // units [cp0, cp0 + ncolpairs)\n"
         "    const u32* __restrict__ lds_image, const v4* __restrict__ table, u32 max_mm,\n"
         "    qent* __restrict__ queue, u64 qcap, u64* __restrict__ qcount, u32* __restrict__ work, u64* __restrict__ stamps,\n"
         "    const u32* __restrict__ rst, const dpat* __restrict__ pats, const u64* __restrict__ rec_start, const u64* __restrict__ rec_len,\n"
         "    const u32* __restrict__ block_rec, u32 nrec, u32 check_rst, hitrec* __restrict__ hits, u64 hcap, u64* __restrict__ counts,\n"
         "    u64* __restrict__ next_counts, u64* __restrict__ next_qcount, u64* __restrict__ pub, hitrec* __restrict__ pub_hits, u32 pre,\n"
         "    u32* __restrict__ pub_seq, u32 seq) {\n"
         "  __shared__ u32 __attribute__((aligned(16))) lds[((LDS_WORDS + 3u) & ~3u) + " << IPCR_INDEX_WAVES << "u * QCAP * 4u]; // static: every LDS address is a compile-time offset\n"
         "  __shared__ u32 wg_cand; // windows this workgroup's exact checks passed (fused form)\n"
         "  if (threadIdx.x == 0u) wg_cand = 0u;\n"
         "  // the counters alternate between two sets; workgroup 0 clears the set the NEXT scan will use\n"
         "  if (blockIdx.x == 0u && next_counts) {\n"
         "    if (threadIdx.x < 4u) next_counts[threadIdx.x] = 0ull;\n"
         "    if (threadIdx.x < 256u) next_qcount[threadIdx.x * 16u] = 0ull;\n"
         "  }\n"
         "  const fuse fu = {rst, pats, rec_start, rec_len, block_rec, nrec, check_rst, hits, hcap, counts, pub_hits, pre, seq};\n"
         "  u32 ncand = 0u;\n"
         "  // the image into LDS, 16 bytes per lane and load (TAB_WORD0 is a multiple of 4; a chunk's launch is one unit per wave: the\n"
         "  // staging is a fifth of its life)\n"
         "  for (u32 i = threadIdx.x; i < TAB_WORD0 / 4u; i += blockDim.x) reinterpret_cast<v4*>(lds)[i] = reinterpret_cast<const v4*>(lds_image)[i];\n"
         "  if (threadIdx.x < 8u) lds[TAB_WORD0 + threadIdx.x] = reinterpret_cast<const u32*>(BITTAB)[threadIdx.x];\n"
         "  __syncthreads();\n"
         "  const u64* T64 = reinterpret_cast<const u64*>(lds);\n"
         "  const unsigned char* ldsb = reinterpret_cast<const unsigned char*>(lds);\n"
         "  const unsigned short* prefix = reinterpret_cast<const unsigned short*>(lds + PREFIX_WORD0);\n"
         "  const u32 lane = threadIdx.x & 63u;\n"
         "  u32* wq = lds + ((LDS_WORDS + 3u) & ~3u) + (threadIdx.x >> 6) * (QCAP * 4u); // this wave's hit queue\n"
         "  u32 qn = 0; // entries queued (wave-uniform)\n"
         "  const u64 wave0 = (u64)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);\n"
         "  const u64 nwaves = (u64)gridDim.x * (blockDim.x >> 6);\n"
         "  if (stamps && lane == 0u) stamps[wave0 * 2u] = __builtin_amdgcn_s_memrealtime();\n"
         "  // the transpose's per-lane constants, and which word of a chunk this lane loads (tile_layout.h: word ((row quad * 3 + plane) * 64\n"
         "  // + column) * 4 + row % 4 of a block; the column part is in the unit's base pointer, a chunk is 8 row quads = 6144 words)\n"
         "  trc tc;\n"
         "  { const u32 j = lane & 31u;\n"
         "    const u32 m0[5] = {0x0000FFFFu, 0x00FF00FFu, 0x0F0F0F0Fu, 0x33333333u, 0x55555555u};\n"
         "    for (int st = 0; st < 5; ++st) { const u32 d = 16u >> st; tc.sh[st] = (j & d) ? d : 32u - d; tc.mk[st] = (j & d) ? ~m0[st] : m0[st]; } }\n"
         "  const u32 jh = lane & 31u;\n"
         "  const u32 rowA = 15u - (jh >> 1), rowB = 31u - (jh >> 1), rowI = 31u - jh;\n"
         "  const u32 offA = ((rowA >> 2) * 3u + (jh & 1u)) * 256u + (rowA & 3u);\n"
         "  const u32 offB = ((rowB >> 2) * 3u + (jh & 1u)) * 256u + (rowB & 3u);\n"
         "  const u32 offI = ((rowI >> 2) * 3u + 2u) * 256u + (rowI & 3u);\n";
    const bool dynamic = env_int("IPCR_INDEX_DYNAMIC", 1, 0, 1) != 0;
    // Which counter?  A 128-byte line of the tiles holds four rows of EIGHT columns, a unit reads two of them: four
    // consecutive units share every line they load.  With one counter for the chip those four go to whichever waves ask
    // next -- usually on four different XCDs, each with an L2 of its own, and the line is fetched four times (PMC: 1.7-3.4 x
    // the algorithmic bytes).  So the groups of four units are dealt to the XCDs in turn, one counter per XCD (work[8 +
    // 32 x]), and a wave asks the counter of the XCD it runs on (HW_REG_XCC_ID: speed only -- every unit is taken exactly
    // once whatever the placement); a wave whose counter has run out moves on to the next one, so the sweep still ends
    // with every wave busy.  IPCR_INDEX_XCD=0: the one counter.
    const bool per_xcd = dynamic && env_int("IPCR_INDEX_XCD", 1, 0, 1) != 0;
    // The next unit is taken, and its first nine loads issued, under the walk of the current unit's last chunk
    // (IPCR_INDEX_AHEAD=0: at the unit's start, where the wave waits for the counter and then for the loads).
    const bool ahead = per_xcd && env_int("IPCR_INDEX_AHEAD", 1, 0, 1) != 0;
    if (per_xcd) {
        s << "  u32 xq = (u32)__builtin_amdgcn_s_getreg(20 | (3 << 11)) & 7u, xtries = 0u; // HW_REG_XCC_ID, bits 3:0\n"
             "  auto take_unit = [&]() __attribute__((always_inline)) { // -> a unit of [0, ncolpairs), or something beyond: no unit is left\n"
             "    u64 unit;\n"
             "    for (;;) {\n"
             "      u32 take = 0u;\n"
             "      if (lane == 0u) take = atomicAdd(work + 8u + 32u * xq, 1u);\n"
             "      take = (u32)__builtin_amdgcn_readfirstlane((int)take);\n"
             "      unit = ((u64)(take >> 2) * 8u + xq) * 4u + (take & 3u); // ascending in take: a counter that has run out stays so\n"
             "      if (unit < ncolpairs || ++xtries == 8u) break;\n"
             "      xq = (xq + 1u) & 7u;\n"
             "    }\n"
             "    return unit;\n"
             "  };\n";
        if (ahead)
            s << "  // a unit's first words as loaded: its last chunk (the history of lane + 1), its first chunk, bit 31 of the column before\n"
                 "  u32 nSA = 0u, nSB = 0u, nSI = 0u, nA = 0u, nB = 0u, nI = 0u, nH0 = 0u, nH1 = 0u;\n"
                 "  auto issue_unit = [&](u64 unit) __attribute__((always_inline)) {\n"
                 "    const u64 ncp = cp0 + unit, ncol = ncp * 2u + (lane >> 5);\n"
                 "    const u32* const nbase = planes + (((ncol >> 6) * 6144u + (ncol & 63u)) << 2);\n"
                 "    nSA = nbase[offA + 18432u]; nSB = nbase[offB + 18432u]; nSI = nbase[offI + 18432u];\n"
                 "    nA = nbase[offA]; nB = nbase[offB]; nI = nbase[offI];\n"
                 "    if (ncp != 0ull) {\n"
                 "      const u64 pcol = ncp * 2u - 1u;\n"
                 "      const u32* const pbase = planes + (((pcol >> 6) * 6144u + (pcol & 63u)) << 2);\n"
                 "      nH0 = pbase[(lane < 32u ? offA : offB) + 18432u]; nH1 = pbase[offI + 18432u];\n"
                 "    }\n"
                 "  };\n"
                 "  u64 nunit = take_unit();\n"
                 "  if (nunit < ncolpairs) issue_unit(nunit);\n"
                 "  for (;;) { // one unit = one column pair = 64 strands\n"
                 "    if (nunit >= ncolpairs) break;\n"
                 "    const u64 cp = cp0 + nunit;\n";
        else
            s << "  for (;;) { // one unit = one column pair = 64 strands\n"
                 "    const u64 unit = take_unit();\n"
                 "    if (unit >= ncolpairs) break;\n"
                 "    const u64 cp = cp0 + unit;\n";
    }
    else if (dynamic)
        // Units are handed out by a counter (work[0]): the waves of a persistent grid do not all run at the same pace (with a
        // fixed share each, the 16 waves of a CU ended between 67 % and 100 % of the sweep: 7.29 ms per 3 Gb, 6.30 ms with the
        // counter).  The last wave to leave zeroes the counters again (work[32] counts the leavers): nothing to clear between launches.
        s << "  for (;;) { // one unit = one column pair = 64 strands\n"
             "    u32 take = 0u;\n"
             "    if (lane == 0u) take = atomicAdd(work, 1u);\n"
             "    if ((u64)(u32)__builtin_amdgcn_readfirstlane((int)take) >= ncolpairs) break;\n"
             "    const u64 cp = cp0 + (u64)(u32)__builtin_amdgcn_readfirstlane((int)take);\n";
    else
        s << "  for (u64 cp = cp0 + wave0; cp < cp0 + ncolpairs; cp += nwaves) { // one unit = one column pair = 64 strands\n";
    // The drain.  The queue is a STACK: a round takes the newest 64 entries, and what it hands back (an entry's further key
    // hits, a key's further patterns) goes into the slots it has just read, on top of the older entries -- so the next round
    // is again 64 entries wide.  A drain in the middle of a unit stops below 64 entries (they wait for company); only the
    // unit's last drain runs rounds that are not full.  (Until late in round 3 the queue was emptied front to back at every
    // drain and the handed-back entries -- a fifth of a pass -- in passes of their own, at a fifth of the lanes and less:
    // 0.23 rounds per base step where the hits fill 0.13.  IPCR_INDEX_STACK_DRAIN=0 is that form.)
    const bool stack_drain = env_int("IPCR_INDEX_STACK_DRAIN", 1, 0, 1) != 0;
    if (stack_drain) {
    s << "    auto flush = [&](bool all) __attribute__((always_inline)) {\n"
         "      u32 n = qn;\n"
         "      const u64 unit_base = cp * 8192u; // first position of this unit\n"
         "      const u32 shard = (u32)cp & 255u;\n"
         "      const u32 floor_n = all ? 0u : 63u;\n"
         "      while (n > floor_n) {\n"
         "        const u32 qb = n > 64u ? n - 64u : 0u; // the round: entries [qb, n)\n"
         "        const u32 i = qb + lane;\n"
         "        v4 e = *reinterpret_cast<const v4*>(wq + i * 4u); // (slots up to 63 are inside the queue whatever n is)\n"
         "        if (i >= n) e.w = 0u; // nothing pending, no chain: the lane idles through the round\n"
         "        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, \"wavefront\"); // every lane has read its slot: slots from qb up may be rewritten\n"
         "        u32 pend = 0u, idx = 0xFFFFFFFFu, back = 0u, keep = 0u; // keep: what of w a handed-back entry keeps (layout C: lane and row)\n"
         "        u32 elane, erow0;\n"
         "        u64 hkm; u32 hbad;\n"
         "        if (EMODE == 0) { hkm = ((u64)(KMHI ? (e.y & ((1u << KMHI) - 1u)) : 0u) << 32) | e.x; hbad = e.z; elane = (e.y >> KMHI) & 63u; erow0 = (e.y >> (KMHI + 6u)) & ((1u << ROWB) - 1u); }\n"
         "        else if (EMODE == 1) { hkm = ((u64)(KMHI ? (e.y & ((1u << KMHI) - 1u)) : 0u) << 32) | e.x; hbad = e.z & ((1u << NBAS) - 1u); elane = e.z >> NBAS; erow0 = (e.y >> KMHI) & ((1u << ROWB) - 1u); }\n"
         "        else { hkm = ((u64)e.y << 32) | e.x; hbad = e.z; elane = (e.w >> 15) & 63u; erow0 = (e.w >> 21) & ((1u << ROWB) - 1u); keep = e.w & 0x7FFF8000u; }\n"
         "        const u32 pay = EMODE == 2 ? (e.w & 0x7FFFu) : (e.w & 0x7FFFFFFFu);\n"
         "        if (e.w & 0x80000000u) { idx = pay >> 2; back = pay & 3u; }\n"
         "        else pend = pay;\n"
         "        const u32 rest = pend & (pend - 1u);\n"
         "        if (pend != 0u) {\n"
         "          const u32 t = reinterpret_cast<const unsigned char*>(lds + TAB_WORD0)[__builtin_ctz(pend)];\n"
         "          const u32 sidx = t & 15u;\n"
         "          back = t >> 4;\n"
         "          const u64 skm = hkm >> (2u * back); // a hit of an earlier step of the entry: its own k-mer\n"
         "          const v4 sc = *reinterpret_cast<const v4*>(lds + SHAPE_WORD0 + 4u * sidx); // {shifts, masks, first bitmap word, first entry}\n"
         "          const u32 key = ((u32)(skm >> (sc.x & 63u)) & (sc.y & 0xFFFFu)) | (((u32)(skm >> ((sc.x >> 8) & 63u)) & (sc.y >> 16)) << ((sc.x >> 16) & 31u));\n"
         "          // the key is in the panel; its rank among the shape's keys is the index of its entry\n"
         "          if (PAIRED) { // keys are ranked as the OLDER step of a pair files them: word = the key's low 4 + 8 bits, bit = its high 2 + 2\n"
         "            const u32 dw = sc.z * 2u + (((key >> 2) & 0xFF0u) | (key & 15u)), bo = ((key >> 14) << 2) | ((key >> 4) & 3u);\n"
         "            const v4 g4 = *reinterpret_cast<const v4*>(lds + (dw & ~3u)); // the 64 keys (four words' low halves) one prefix covers\n"
         "            const u32 pos = dw & 3u;\n"
         "            const u32 own = pos == 0u ? g4.x : (pos == 1u ? g4.y : (pos == 2u ? g4.z : g4.w));\n"
         "            u32 r = (u32)__builtin_popcount(own & ((1u << bo) - 1u));\n"
         "            if (pos > 0u) r += (u32)__builtin_popcount(g4.x & 0xFFFFu);\n"
         "            if (pos > 1u) r += (u32)__builtin_popcount(g4.y & 0xFFFFu);\n"
         "            if (pos > 2u) r += (u32)__builtin_popcount(g4.z & 0xFFFFu);\n"
         "            idx = sc.w + (u32)prefix[dw >> 2] + r;\n"
         "          } else {\n"
         "            const u32 wi = sc.z + (key >> 6); // the shape's bitmap word with this key\n"
         "            const u64 w = T64[wi];\n"
         "            idx = sc.w + (u32)prefix[wi] + (u32)__popcll((w << (63u - (key & 63u))) << 1);\n"
         "          }\n"
         "        }\n"
         "        u32 next = 0xFFFFFFFFu;\n"
         "        if (idx != 0xFFFFFFFFu) {\n"
         "          const u64 skm = hkm >> (2u * back);\n"
         "          const u32 sbad = hbad >> back;\n"
         "          const u32 strand_off = elane << 7;\n"
         "          const int erow = (int)erow0 - (int)back;\n"
         "          next = check_entry(idx, skm, sbad, erow, unit_base, strand_off, shard, table, max_mm, queue, qcap, qcount, fu, ncand);\n"
         "          if (!CHAIN_CARRY || rest != 0u)\n"
         "            while (next != 0xFFFFFFFFu) next = check_entry(next, skm, sbad, erow, unit_base, strand_off, shard, table, max_mm, queue, qcap, qcount, fu, ncand);\n"
         "        }\n"
         "        // a lane hands back at most one entry per round: the rest of its mask (any chain under the key it took was\n"
         "        // walked above), or the next pattern of its key\n"
         "        const bool again = rest != 0u || (CHAIN_CARRY && next != 0xFFFFFFFFu);\n"
         "        const u64 rb = __ballot(again);\n"
         "        if (again) {\n"
         "          const u32 slot = qb + __builtin_amdgcn_mbcnt_hi((u32)(rb >> 32), __builtin_amdgcn_mbcnt_lo((u32)rb, 0u));\n"
         "          e.w = rest != 0u ? (keep | rest) : (keep | 0x80000000u | (next << 2) | back); // (k-mer, flags and row as filed)\n"
         "          *reinterpret_cast<v4*>(wq + slot * 4u) = e;\n"
         "        }\n"
         "        n = qb + (u32)__popcll(rb);\n"
         "        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, \"wavefront\"); // the entries handed back are read by other lanes next\n"
         "      }\n"
         "      qn = n;\n"
         "    };\n";
    } else
    s << "    auto flush = [&](bool) __attribute__((always_inline)) {\n"
         "      u32 n = qn;\n"
         "      const u64 unit_base = cp * 8192u; // first position of this unit\n"
         "      const u32 shard = (u32)cp & 255u;\n"
         "      while (n != 0u) {\n"
         "        u32 nc = 0u; // entries handed back so far: slots [0, nc), always behind the round being read\n"
         "        auto hand_back = [&](bool mine, v4 e, u32 w) __attribute__((always_inline)) {\n"
         "          const u64 rb = __ballot(mine);\n"
         "          if (rb != 0ull) {\n"
         "            if (mine) {\n"
         "              const u32 slot = nc + __builtin_amdgcn_mbcnt_hi((u32)(rb >> 32), __builtin_amdgcn_mbcnt_lo((u32)rb, 0u));\n"
         "              e.w = w;\n"
         "              *reinterpret_cast<v4*>(wq + slot * 4u) = e;\n"
         "            }\n"
         "            nc += (u32)__popcll(rb);\n"
         "          }\n"
         "        };\n"
         "        for (u32 qb = 0; qb < n; qb += 64u) {\n"
         "          const u32 i = qb + lane;\n"
         "          v4 e = *reinterpret_cast<const v4*>(wq + i * 4u); // (a slot behind the last entry is still inside the queue: n <= QCAP, a multiple of 64)\n"
         "          if (i >= n) e.w = 0u; // nothing pending, no chain: the lane idles through the round\n"
         "          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, \"wavefront\"); // every lane has read its slot: slots up to qb + 63 may be rewritten\n"
         "          u32 pend = 0u, idx = 0xFFFFFFFFu, back = 0u, keep = 0u; // keep: what of w a handed-back entry keeps (layout C: lane and row)\n"
         "          u32 elane, erow0;\n"
         "          u64 hkm; u32 hbad;\n"
         "          if (EMODE == 0) { hkm = ((u64)(KMHI ? (e.y & ((1u << KMHI) - 1u)) : 0u) << 32) | e.x; hbad = e.z; elane = (e.y >> KMHI) & 63u; erow0 = (e.y >> (KMHI + 6u)) & ((1u << ROWB) - 1u); }\n"
         "          else if (EMODE == 1) { hkm = ((u64)(KMHI ? (e.y & ((1u << KMHI) - 1u)) : 0u) << 32) | e.x; hbad = e.z & ((1u << NBAS) - 1u); elane = e.z >> NBAS; erow0 = (e.y >> KMHI) & ((1u << ROWB) - 1u); }\n"
         "          else { hkm = ((u64)e.y << 32) | e.x; hbad = e.z; elane = (e.w >> 15) & 63u; erow0 = (e.w >> 21) & ((1u << ROWB) - 1u); keep = e.w & 0x7FFF8000u; }\n"
         "          const u32 pay = EMODE == 2 ? (e.w & 0x7FFFu) : (e.w & 0x7FFFFFFFu);\n"
         "          if (e.w & 0x80000000u) { idx = pay >> 2; back = pay & 3u; }\n"
         "          else pend = pay;\n"
         "          const u32 rest = pend & (pend - 1u);\n"
         "          hand_back(rest != 0u, e, keep | rest); // (k-mer, flags and row as filed)\n"
         "          if (pend != 0u) {\n"
         "            const u32 t = reinterpret_cast<const unsigned char*>(lds + TAB_WORD0)[__builtin_ctz(pend)];\n"
         "            const u32 sidx = t & 15u;\n"
         "            back = t >> 4;\n"
         "            const u64 skm = hkm >> (2u * back); // a hit of an earlier step of the entry: its own k-mer\n"
         "            const v4 sc = *reinterpret_cast<const v4*>(lds + SHAPE_WORD0 + 4u * sidx); // {shifts, masks, first bitmap word, first entry}\n"
         "            const u32 key = ((u32)(skm >> (sc.x & 63u)) & (sc.y & 0xFFFFu)) | (((u32)(skm >> ((sc.x >> 8) & 63u)) & (sc.y >> 16)) << ((sc.x >> 16) & 31u));\n"
         "            // the key is in the panel; its rank among the shape's keys is the index of its entry\n"
         "            if (PAIRED) { // keys are ranked as the OLDER step of a pair files them: word = the key's low 4 + 8 bits, bit = its high 2 + 2\n"
         "              const u32 dw = sc.z * 2u + (((key >> 2) & 0xFF0u) | (key & 15u)), bo = ((key >> 14) << 2) | ((key >> 4) & 3u);\n"
         "              const v4 g4 = *reinterpret_cast<const v4*>(lds + (dw & ~3u)); // the 64 keys (four words' low halves) one prefix covers\n"
         "              const u32 pos = dw & 3u;\n"
         "              const u32 own = pos == 0u ? g4.x : (pos == 1u ? g4.y : (pos == 2u ? g4.z : g4.w));\n"
         "              u32 r = (u32)__builtin_popcount(own & ((1u << bo) - 1u));\n"
         "              if (pos > 0u) r += (u32)__builtin_popcount(g4.x & 0xFFFFu);\n"
         "              if (pos > 1u) r += (u32)__builtin_popcount(g4.y & 0xFFFFu);\n"
         "              if (pos > 2u) r += (u32)__builtin_popcount(g4.z & 0xFFFFu);\n"
         "              idx = sc.w + (u32)prefix[dw >> 2] + r;\n"
         "            } else {\n"
         "              const u32 wi = sc.z + (key >> 6); // the shape's bitmap word with this key\n"
         "              const u64 w = T64[wi];\n"
         "              idx = sc.w + (u32)prefix[wi] + (u32)__popcll((w << (63u - (key & 63u))) << 1);\n"
         "            }\n"
         "          }\n"
         "          u32 next = 0xFFFFFFFFu;\n"
         "          if (idx != 0xFFFFFFFFu) {\n"
         "            const u64 skm = hkm >> (2u * back);\n"
         "            const u32 sbad = hbad >> back;\n"
         "            const u32 strand_off = elane << 7;\n"
         "            const int erow = (int)erow0 - (int)back;\n"
         "            next = check_entry(idx, skm, sbad, erow, unit_base, strand_off, shard, table, max_mm, queue, qcap, qcount, fu, ncand);\n"
         "            if (!CHAIN_CARRY || rest != 0u)\n"
         "              while (next != 0xFFFFFFFFu) next = check_entry(next, skm, sbad, erow, unit_base, strand_off, shard, table, max_mm, queue, qcap, qcount, fu, ncand);\n"
         "          }\n"
         "          if (CHAIN_CARRY) hand_back(next != 0xFFFFFFFFu, e, keep | 0x80000000u | (next << 2) | back);\n"
         "        }\n"
         "        n = nc;\n"
         "        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, \"wavefront\"); // the entries handed back are read by other lanes next\n"
         "      }\n"
         "      qn = 0;\n"
         "      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, \"wavefront\"); // queue slots are rewritten by other lanes next\n"
         "    };\n";
    // the unit's tiles: column 2 cp + (lane >> 5)
    if (ahead)
    s << "    const u64 col = cp * 2u + (lane >> 5);\n"
         "    const u32* const base = planes + (((col >> 6) * 6144u + (col & 63u)) << 2);\n"
         "    // (the words were loaded under the unit before: issue_unit)\n"
         "    const u32 SA = tr32(nSA, tc), SB = tr32(nSB, tc), SI = tr32(nSI, tc);\n"
         "    u32 PA = (u32)__shfl_up((int)SA, 1), PB = (u32)__shfl_up((int)SB, 1), PI = (u32)__shfl_up((int)SI, 1);\n"
         "    {\n"
         "      u32 hA = 0u, hB = 0u, hI = 0xFFFFFFFFu; // the very first strand: nothing but invalid bases in front of it\n"
         "      if (cp != 0ull) {\n"
         "        const u64 bk = __ballot((nH0 >> 31) != 0u);\n"
         "        const u64 bi = __ballot((nH1 >> 31) != 0u);\n"
         "        hA = (u32)bk; hB = (u32)(bk >> 32); hI = (u32)bi;\n"
         "      }\n"
         "      if (lane == 0u) { PA = hA; PB = hB; PI = hI; }\n"
         "    }\n"
         "    u32 A = tr32(nA, tc), B = tr32(nB, tc), I = tr32(nI, tc);\n";
    else
    s << "    const u64 col = cp * 2u + (lane >> 5);\n"
         "    const u32* const base = planes + (((col >> 6) * 6144u + (col & 63u)) << 2);\n"
         "    // history: the strand before mine ended with its chunk 3 -- lane - 1's, which is loaded and transposed first (and kept: the\n"
         "    // walk comes back to it); lane 0's predecessor is strand 31 of the column before the unit's first: bit 31 of that column's\n"
         "    // words, gathered by a ballot over the lanes that load them (half 0: the rows of word A, half 1: those of word B)\n"
         "    const u32 SA = tr32(base[offA + 18432u], tc), SB = tr32(base[offB + 18432u], tc), SI = tr32(base[offI + 18432u], tc);\n"
         "    u32 PA = (u32)__shfl_up((int)SA, 1), PB = (u32)__shfl_up((int)SB, 1), PI = (u32)__shfl_up((int)SI, 1);\n"
         "    {\n"
         "      u32 hA = 0u, hB = 0u, hI = 0xFFFFFFFFu; // the very first strand: nothing but invalid bases in front of it\n"
         "      if (cp != 0ull) {\n"
         "        const u64 pcol = cp * 2u - 1u;\n"
         "        const u32* const pbase = planes + (((pcol >> 6) * 6144u + (pcol & 63u)) << 2);\n"
         "        const u64 bk = __ballot((pbase[(lane < 32u ? offA : offB) + 18432u] >> 31) != 0u);\n"
         "        const u64 bi = __ballot((pbase[offI + 18432u] >> 31) != 0u);\n"
         "        hA = (u32)bk; hB = (u32)(bk >> 32); hI = (u32)bi;\n"
         "      }\n"
         "      if (lane == 0u) { PA = hA; PB = hB; PI = hI; }\n"
         "    }\n"
         "    u32 A = tr32(base[offA], tc), B = tr32(base[offB], tc), I = tr32(base[offI], tc);\n";
    s << "    u32 rA = 0u, rB = 0u, rI = 0u; // the next chunk's words as loaded\n"
         "    u32 acc = 0u; // the masks of the steps since the last queue entry\n";
    for (unsigned j = 0; j < NCH && paired; ++j) s << "    u32 acc" << j << " = 0u;\n";
    if (mode == 0) s << "    const u32 lane_y = lane << KMHI; // this lane's part of an entry's y\n";
    if (mode == 1) s << "    const u32 lane_z = lane << NBAS;\n";
    if (mode == 2) s << "    const u32 lane_w = lane << 15;\n";
    s << "    u32 it = 0u, u = 0u; // chunk, step of the chunk\n"
         "    bool done = false;\n"
         "    while (!done) {\n";
    // A queue entry's SPE steps are ONE block of code (a drain resumes behind an entry).  Their lookups are independent of
    // one another -- every key field is a constant slice of the same three registers.  IPCR_INDEX_PHASED=1 issues all the
    // block's LDS reads before any test (24 in flight for C4): measured 3 % SLOWER than letting every step wait for its
    // own six (the LDS is the busy unit, 81 % of the sweep: bursts only queue up behind one another).
    const bool phased = env_int("IPCR_INDEX_PHASED", 0, 0, 1) != 0;
    for (unsigned k0 = 0; k0 < U; k0 += SPE) {
        std::ostringstream pa, pb; // phase A (addresses, reads), phase B (tests)
        if (paired) {
            unsigned cc = 0; // tests made so far in this block
            auto test = [&](const std::string &word, const std::string &idx) {
                const std::string a = "acc" + std::to_string(cc % NCH);
                if (use_add) pb << "        " << a << " = ANDOR(" << word << " >> " << idx << ", 1u, " << a << " + " << a << ");\n";
                else pb << "        " << a << " = __builtin_amdgcn_alignbit(" << word << " >> " << idx << ", " << a << ", 1u);\n";
                ++cc;
            };
            for (unsigned kn = k0 + 1u; kn < k0 + SPE; kn += 2u) { // the pair (kn - 1, kn), every field taken in the frame of step kn
                const unsigned half = kn / 16u, tq = kn % 16u, P0 = 2u * (15u - tq);
                const char *W[3] = {half ? "B" : "A", half ? "A" : "PB", half ? "PB" : "PA"};
                auto shifted = [&](unsigned f, unsigned width) { // (km >> f), valid in its low `width` bits at least; km = {W2, W1, W0} >> P0
                    const unsigned P = P0 + f, q = P / 32u, r = P % 32u;
                    if (q > 2u) return std::string("0u");
                    if (r == 0u) return std::string(W[q]);
                    if (r + width <= 32u || q == 2u) return "(" + std::string(W[q]) + " >> " + std::to_string(r) + "u)";
                    return "__builtin_amdgcn_alignbit(" + std::string(W[q + 1]) + ", " + W[q] + ", " + std::to_string(r) + "u)";
                };
                auto width_of = [](unsigned mask) { unsigned w = 0; while (mask >> w) ++w; return w; };
                auto field = [&](unsigned f, unsigned mask) { return "(" + shifted(f, width_of(mask)) + " & " + std::to_string(mask) + "u)"; };
                const std::string K = std::to_string(kn);
                pa << "        // steps " << kn - 1u << ", " << kn << "\n";
                for (size_t gi = 0; gi < groups.size(); ++gi) {
                    const unsigned c = (unsigned)groups[gi].c_off; // the protected bases: 8 bits of the two steps together
                    const std::string G = std::to_string(gi) + "_" + K;
                    pa << "        const u32 cp" << G << " = " << field(c, 0x3Cu) << ", un" << G << " = ANDOR(" << shifted(c, 2u) << ", 3u, 16u), uo" << G << " = " << field(c + 6u, 3u) << ";\n";
                    for (int si : groups[gi].sh) {
                        const unsigned b = (unsigned)shapes[(size_t)si].blk_shift; // the block: 12 bits of the two steps together
                        const std::string S = std::to_string(si) + "_" + K;
                        const std::string addr = b >= 4u ? "ANDOR(" + shifted(b - 4u, 14u) + ", 16320u, cp" + G + ")" : "((" + field(b + 2u, 255u) + " << 6) | cp" + G + ")";
                        const std::string inew = b >= 2u ? "ANDOR(" + shifted(b - 2u, 4u) + ", 12u, un" + G + ")" : "((" + field(b, 3u) + " << 2) | un" + G + ")";
                        const std::string iold = "ANDOR(" + shifted(b + 8u, 4u) + ", 12u, uo" + G + ")";
                        pa << "        const u32 w" << S << " = *reinterpret_cast<const u32*>(ldsb + " << off64[(size_t)si] * 8u << "u + " << addr << ");\n";
                        test("w" + S, iold);
                        test("w" + S, inew);
                    }
                }
            }
            if (use_add) {
                pb << "        acc = (acc0 & " << ((1u << NPC) - 1u) << "u)";
                for (unsigned j = 1; j < NCH; ++j) pb << " | ((acc" << j << " & " << ((1u << NPC) - 1u) << "u) << " << j * NPC << "u)";
                pb << ";\n";
            } else {
                pb << "        acc = (acc0 >> " << 32u - NPC << "u)";
                for (unsigned j = 1; j < NCH; ++j) pb << " | ((acc" << j << " >> " << 32u - NPC << "u) << " << j * NPC << "u)";
                pb << ";\n";
            }
        } else
        for (unsigned k = k0; k < k0 + SPE; ++k) {
            const unsigned half = k / 16u, tq = k % 16u, P0 = 2u * (15u - tq);
            const char *W[3] = {half ? "B" : "A", half ? "A" : "PB", half ? "PB" : "PA"};
            auto shifted = [&](unsigned f, unsigned width) { // (km >> f), valid in its low `width` bits at least; km = {W2, W1, W0} >> P0
                const unsigned P = P0 + f, q = P / 32u, r = P % 32u;
                if (q > 2u) return std::string("0u");
                if (r == 0u) return std::string(W[q]);
                if (r + width <= 32u || q == 2u) return "(" + std::string(W[q]) + " >> " + std::to_string(r) + "u)";
                return "__builtin_amdgcn_alignbit(" + std::string(W[q + 1]) + ", " + W[q] + ", " + std::to_string(r) + "u)";
            };
            auto width_of = [](unsigned mask) { unsigned w = 0; while (mask >> w) ++w; return w; };
            auto field = [&](unsigned f, unsigned mask) { return "(" + shifted(f, width_of(mask)) + " & " + std::to_string(mask) + "u)"; };
            const std::string K = std::to_string(k);
            pa << "        // step " << k << "\n";
            std::vector<char> group_c(groups.size(), 0);
            bool first_pack = true;
            for (unsigned p = 0; p < NPK; ++p) {
                const Pack &pk = packs[p];
                std::string q;
                if (pk.fast) {
                    const size_t gi = (size_t)pk.group;
                    const std::string G = std::to_string(gi) + "_" + K;
                    if (!group_c[gi]) {
                        pa << "        const u32 cb" << G << " = " << field((unsigned)groups[gi].c_off, 7u) << ", cw" << G << " = " << field((unsigned)groups[gi].c_off + 3u, 7u) << ";\n";
                        group_c[gi] = 1;
                    }
                    std::vector<std::string> bytes;
                    for (size_t i = 0; i < pk.sh.size(); ++i) {
                        const int si = pk.sh[i];
                        const ipcr_index_shape &sh = shapes[(size_t)si];
                        // byte address = (bitmap word index) * 8 + the high three bits of c.
                        // A one-shape group keys on protected bases only: the word index is what follows the six bits of c
                        const bool single = sh.blk_mask == 0;
                        const unsigned off = single ? (unsigned)sh.tw_shift + 6u : (unsigned)sh.blk_shift;
                        const unsigned vmask = single ? ((1u << (sh.tw_bits - 6)) - 1u) : sh.blk_mask;
                        const std::string cw = "cw" + G;
                        std::string a;
                        if (vmask == 0) a = cw;
                        else if (off == (unsigned)groups[gi].c_off + 6u) a = field(off - 3u, (vmask << 3) | 7u); // the block lies right above the protected bases: one slice of the k-mer
                        else if (off >= 3) a = "ANDOR(" + shifted(off - 3u, width_of(vmask) + 3u) + ", " + std::to_string(vmask << 3) + "u, " + cw + ")";
                        else a = "((" + field(off, vmask) + " << 3) | " + cw + ")";
                        const std::string nm = "y" + std::to_string(si) + "_" + K;
                        pa << "        const u32 " << nm << " = ldsb[" << off64[(size_t)si] * 8u << "u + " << a << "];\n";
                        bytes.push_back(nm);
                    }
                    std::string packed = bytes.back(); // ((b2 << 8 | b1) << 8) | b0: one v_lshl_or_b32 per byte (left alone the compiler shifts each byte and ORs three)
                    for (size_t i = bytes.size() - 1; i-- > 0;) packed = "LSHL_OR(" + packed + ", 8u, " + bytes[i] + ")";
                    unsigned m = 0;
                    for (size_t i = 0; i < pk.sh.size(); ++i) m |= 1u << (8 * i);
                    q = "((" + packed + " >> cb" + G + ") & " + std::to_string(m) + "u)";
                } else {
                    const int si = pk.sh[0];
                    const ipcr_index_shape &sh = shapes[(size_t)si];
                    std::string key = sh.tw_mask ? field(sh.tw_shift, sh.tw_mask) : std::string("0u");
                    if (sh.blk_mask) key = "(" + key + " | (" + field(sh.blk_shift, sh.blk_mask) + " << " + std::to_string(sh.tw_bits) + "u))";
                    const std::string kn = "key" + std::to_string(si) + "_" + K, wn = "w" + std::to_string(si) + "_" + K;
                    pa << "        const u32 " << kn << " = " << key << ";\n";
                    pa << "        const u32 " << wn << " = lds[" << off64[(size_t)si] * 2u << "u + (" << kn << " >> 5)];\n";
                    q = "__builtin_amdgcn_ubfe(" + wn + ", " + kn + ", 1u)"; // v_bfe_u32 takes the low five bits of the offset itself: one instruction for and, shift, and
                }
                if (first_pack) pb << "        u32 hm" << K << " = " << q << ";\n";
                else // (kept apart: the compiler would otherwise pull the shift in front of the mask and spend a second mask on it)
                    pb << "        { u32 qq = " << q << "; asm volatile(\"\" : \"+v\"(qq)); hm" << K << " = (qq << " << p << "u) | hm" << K << "; }\n";
                first_pack = false;
            }
            if (SPE == 1u || k % SPE == 0u) pb << "        acc = hm" << K << ";\n";
            else pb << "        acc = (acc << " << SHF << "u) | hm" << K << ";\n";
        }
        const unsigned k = k0 + SPE - 1u; // the step that files the entry
        const unsigned half = k / 16u, tq = k % 16u, P0 = 2u * (15u - tq);
        const char *W[3] = {half ? "B" : "A", half ? "A" : "PB", half ? "PB" : "PA"};
        s << "      if (u == " << k0 << "u) {\n";
        if (k0 == 0) {
            s << "        if (it < 2u) { rA = base[offA + (it + 1u) * 6144u]; rB = base[offB + (it + 1u) * 6144u]; rI = base[offI + (it + 1u) * 6144u]; }\n";
            if (ahead)
                s << "        if (it == 3u) { nunit = take_unit(); if (nunit < ncolpairs) issue_unit(nunit); } // the next unit, under this chunk's walk\n";
        }
        s << pa.str();
        if (phased) s << "        __builtin_amdgcn_sched_barrier(0);\n";
        s << pb.str();
        {
            const std::string kmlo = P0 ? "__builtin_amdgcn_alignbit(" + std::string(W[1]) + ", " + W[0] + ", " + std::to_string(P0) + "u)" : std::string(W[0]);
            const std::string kmhi = P0 ? "__builtin_amdgcn_alignbit(" + std::string(W[2]) + ", " + W[1] + ", " + std::to_string(P0) + "u)" : std::string(W[1]);
            const std::string bad = k == 31u ? std::string("I") : "__builtin_amdgcn_alignbit(PI, I, " + std::to_string(31u - k) + "u)";
            s << "        const u64 bal = __ballot(acc != 0u);\n"
                 "        if (bal != 0ull) { // one queue entry per lane whatever the number of shapes and steps that hit\n"
                 "          if (acc != 0u) {\n"
                 "            const u32 slot = qn + __builtin_amdgcn_mbcnt_hi((u32)(bal >> 32), __builtin_amdgcn_mbcnt_lo((u32)bal, 0u));\n"
                 "            const u32 row = it * 32u + " << k << "u;\n"
                 "            v4 e; e.x = " << kmlo << ";\n";
            if (mode == 0)
                s << "            e.y = (KMHI ? (" << kmhi << " & ((1u << KMHI) - 1u)) : 0u) | lane_y | (row << (KMHI + 6u)); e.z = " << bad << "; e.w = acc;\n";
            else if (mode == 1)
                s << "            e.y = (KMHI ? (" << kmhi << " & ((1u << KMHI) - 1u)) : 0u) | (row << KMHI); e.z = (" << bad << " & ((1u << NBAS) - 1u)) | lane_z; e.w = acc;\n";
            else
                s << "            e.y = " << kmhi << "; e.z = " << bad << "; e.w = acc | lane_w | (row << 21);\n";
            s << "            *reinterpret_cast<v4*>(wq + slot * 4u) = e;\n"
                 "          }\n"
                 "          qn += (u32)__popcll(bal);\n"
                 "        }\n";
        }
        s << "        u = " << k + 1 << "u;\n";
        s << "        if (qn > QCAP - 64u) u |= 256u;\n";
        s << "      }\n";
    }
    s << "      const bool full = (u & 256u) != 0u;\n"
         "      u &= 255u;\n"
         "      if (u == " << U << "u) { // the chunk is walked: the next one's words take its place\n"
         "        u = 0u; ++it;\n"
         "        if (it == 4u) done = true;\n"
         "        else { PA = A; PB = B; PI = I; if (it == 3u) { A = SA; B = SB; I = SI; } else { A = tr32(rA, tc); B = tr32(rB, tc); I = tr32(rI, tc); } }\n"
         "      }\n"
         "      if (full || done) { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, \"wavefront\"); flush(done); }\n"
         "    }\n";
    s << "  }\n"; // (rows are relative to the unit: the queue is always empty when a unit ends)
    if (dynamic)
        s << "  if (hits) { // fused: the candidate statistics (one atomic per workgroup), and every record of this wave on its way before it leaves\n"
             "    u32 cs = ncand;\n"
             "    for (int off = 32; off > 0; off >>= 1) cs += __shfl_down(cs, off);\n"
             "    if (lane == 0u && cs != 0u) atomicAdd(&wg_cand, cs);\n"
             "    asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n"
             "    __syncthreads();\n"
             "    if (threadIdx.x == 0u && wg_cand != 0u) { atomicAdd(counts + 2, (u64)wg_cand); asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\"); }\n"
             "  }\n"
             "  u32 last = 0u;\n"
             "  if (lane == 0u) {\n"
             "    const u32 left = atomicAdd(work + 32u, 1u);\n"
             "    if ((u64)left + 1ull == nwaves) {\n"
             "      last = 1u;\n"
             "      __hip_atomic_store(work, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(work + 32u, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);\n"
          << (per_xcd ? "      for (u32 x = 0; x < 8u; ++x) __hip_atomic_store(work + 8u + 32u * x, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);\n" : "") <<
             "    }\n"
             "  }\n"
             "  if (__builtin_amdgcn_readfirstlane((int)last) != 0 && pub) { // every other wave of the scan has left: hand the counters to the host\n"
             "    __threadfence();\n"
             "    if (lane < 4u) pub[lane] = __hip_atomic_load(counts + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);\n"
             "    __threadfence_system();\n"
             "    if (lane == 0u) __hip_atomic_store(pub_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);\n"
             "  }\n";
    s << "  if (stamps && lane == 0u) stamps[wave0 * 2u + 1u] = __builtin_amdgcn_s_memrealtime();\n";
    s << "}\n";
    return s.str();
}

JitFilter *jit_build_index(const std::vector<ipcr_index_shape> &shapes, const IndexGeom &geom, std::string &err) {
    if (shapes.empty() || shapes.size() > IPCR_INDEX_MAX_SHAPES) { err = "no index shapes"; return nullptr; }
    int dev = 0;
    (void)hipGetDevice(&dev);
    hipDeviceProp_t prop;
    std::string arch = "gfx950";
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.gcnArchName[0]) {
        arch = prop.gcnArchName;
        const size_t colon = arch.find(':');
        if (colon != std::string::npos) arch = arch.substr(0, colon);
    }
    std::vector<char> code;
    if (!compile_group(jit_index_source(shapes, geom), arch, code, err)) return nullptr;
    JitFilter *f = new JitFilter;
    f->waves_per_group = IPCR_INDEX_WAVES;
    if (hipModuleLoadData(&f->module, code.data()) != hipSuccess ||
        hipModuleGetFunction(&f->fn, f->module, "ipcr_index_filter") != hipSuccess) {
        err = "hipModuleLoadData/GetFunction failed for the specialised index filter";
        jit_destroy(f);
        return nullptr;
    }
    return f; // the kernel's LDS (image + hit queues, up to 160 KiB) is static: nothing to request at launch
}

bool jit_index_fusable() { return env_int("IPCR_INDEX_DYNAMIC", 1, 0, 1) != 0 && env_int("IPCR_INDEX_FUSED", 1, 0, 1) != 0; }

hipError_t jit_launch_index(JitFilter *f, hipStream_t st, const uint32_t *planes, uint64_t block0, uint64_t nblocks, uint32_t /*nshapes*/,
                            const uint32_t *lds_image, const void *table, uint32_t max_mm, void *queue,
                            uint64_t qcap, unsigned long long *qcount, uint32_t *work, hipEvent_t start, hipEvent_t stop, const JitVerify *fused) {
    if (nblocks == 0) return hipSuccess;
    uint64_t cp0 = block0 * 32u, ncolpairs = nblocks * 32u; // one unit = one column pair
    // one persistent 16-wave workgroup per CU: the bitmaps are staged into LDS once per CU
    uint64_t grid = 256ull;
    if (grid * IPCR_INDEX_WAVES > ncolpairs) grid = (ncolpairs + IPCR_INDEX_WAVES - 1u) / IPCR_INDEX_WAVES;
    // dev tool: IPCR_INDEX_STAMPS=<file> appends every wave's start / end time (100 MHz counter) of every sweep
    static const char *stamp_path = getenv("IPCR_INDEX_STAMPS");
    unsigned long long *stamps = nullptr;
    const size_t nstamp = (size_t)grid * IPCR_INDEX_WAVES * 2u;
    if (stamp_path && *stamp_path) {
        if (hipMalloc((void **)&stamps, nstamp * 8u) != hipSuccess) stamps = nullptr;
        else (void)hipMemsetAsync(stamps, 0, nstamp * 8u, st);
    }
    JitVerify a = fused ? *fused : JitVerify(); // (all null without: survivors go to the candidate queue, the stand-alone verifier follows)
    void *args[] = {(void *)&planes, (void *)&cp0, (void *)&ncolpairs, (void *)&lds_image, (void *)&table,
                    (void *)&max_mm, (void *)&queue, (void *)&qcap, (void *)&qcount, (void *)&work, (void *)&stamps,
                    (void *)&a.rst, (void *)&a.pats, (void *)&a.rec_start, (void *)&a.rec_len, (void *)&a.block_rec, (void *)&a.nrec, (void *)&a.check_rst,
                    (void *)&a.hits, (void *)&a.hcap, (void *)&a.counts, (void *)&a.next_counts, (void *)&a.next_qcount, (void *)&a.pub,
                    (void *)&a.pub_hits, (void *)&a.pre, (void *)&a.pub_seq, (void *)&a.seq};
    const hipError_t e = hipExtModuleLaunchKernel(f->fn, (unsigned)grid * IPCR_INDEX_WAVES * 64u, 1, 1, IPCR_INDEX_WAVES * 64u, 1, 1, 0, st, args, nullptr, start, stop, 0);
    if (stamps) {
        std::vector<unsigned long long> h(nstamp);
        if (e == hipSuccess && hipStreamSynchronize(st) == hipSuccess && hipMemcpy(h.data(), stamps, nstamp * 8u, hipMemcpyDeviceToHost) == hipSuccess)
            if (FILE *fh = fopen(stamp_path, "ab")) {
                const unsigned long long n = nstamp;
                fwrite(&n, 8, 1, fh);
                fwrite(h.data(), 8, nstamp, fh);
                fclose(fh);
            }
        (void)hipFree(stamps);
    }
    return e;
}

namespace {
std::atomic<uint64_t> g_small_launches{0};

// builds f's form for small launches (f->small); false when it cannot be had
bool build_small(JitFilter *f, int segs) {
    const std::string src = jit_source(f->pats, f->max_mm, f->qbase, f->has_ids ? &f->ids : nullptr, f->spill_only, segs);
    if (src.empty()) return false;
    std::vector<char> code;
    std::string err;
    if (!compile_group(src, f->arch, code, err)) return false;
    JitFilter *sm = new JitFilter;
    sm->segments = (unsigned)segs;
    const size_t at = src.find("// IPCR_WAVES_PER_GROUP ");
    if (at != std::string::npos) sm->waves_per_group = (unsigned)atoi(src.c_str() + at + 24);
    if (hipModuleLoadData(&sm->module, code.data()) != hipSuccess || hipModuleGetFunction(&sm->fn, sm->module, "ipcr_filter") != hipSuccess) {
        jit_destroy(sm);
        return false;
    }
    f->small = sm;
    return true;
}

// the segmented form of f if it is there; the first caller to ask starts its build (in the background unless IPCR_JIT_ASYNC=0)
JitFilter *small_form(JitFilter *f, int segs) {
    int st = f->small_state.load(std::memory_order_acquire);
    if (st == 2) return f->small;
    if (st != 0) return nullptr;
    int expected = 0;
    if (!f->small_state.compare_exchange_strong(expected, 1)) return f->small_state.load(std::memory_order_acquire) == 2 ? f->small : nullptr;
    static const bool async_on = !(getenv("IPCR_JIT_ASYNC") && atoi(getenv("IPCR_JIT_ASYNC")) == 0);
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!async_on) {
        f->small_state.store(build_small(f, segs) ? 2 : 3, std::memory_order_release);
        return f->small_state.load() == 2 ? f->small : nullptr;
    }
    f->small_thread = std::thread([f, segs, dev] {
        (void)hipSetDevice(dev); // the code object is loaded onto the device of the launch that asked
        f->small_state.store(build_small(f, segs) ? 2 : 3, std::memory_order_release);
    });
    return nullptr;
}
} // namespace

hipError_t jit_launch(JitFilter *f, hipStream_t st, const uint32_t *planes, uint64_t block0, uint64_t nblocks, void *queue,
                      uint64_t qcap, unsigned long long *qcount, const JitVerify &v, hipEvent_t start, hipEvent_t stop) {
    if (nblocks == 0) return hipSuccess;
    JitVerify a = v;
    void *args[] = {(void *)&planes, (void *)&block0, (void *)&nblocks, (void *)&queue, (void *)&qcap, (void *)&qcount,
                    (void *)&a.rst, (void *)&a.pats, (void *)&a.rec_start, (void *)&a.rec_len, (void *)&a.block_rec, (void *)&a.nrec,
                    (void *)&a.max_mm, (void *)&a.check_rst, (void *)&a.hits, (void *)&a.hcap, (void *)&a.counts,
                    (void *)&a.next_counts, (void *)&a.next_qcount, (void *)&a.tickets, (void *)&a.pub,
                    (void *)&a.pub_hits, (void *)&a.pre, (void *)&a.pub_seq, (void *)&a.seq, (void *)&a.withhold};
    // A small launch -- a chunk of the drop-in path, a bacterial genome -- takes the form in which a block is shared by several
    // waves, as soon as that has been built (IPCR_JIT_SEGMENTS=1: never; IPCR_JIT_SEG_BLOCKS: what "small" is)
    const int seg_n = env_int("IPCR_JIT_SEGMENTS", 4, 1, 8); // (read per launch: the tests run both forms in one process)
    const int seg_blocks = env_int("IPCR_JIT_SEG_BLOCKS", 512, 0, 1 << 20);
    if (seg_n > 1 && nblocks <= (uint64_t)seg_blocks && f->segments == 1)
        if (JitFilter *sm = small_form(f, seg_n)) { f = sm; g_small_launches.fetch_add(1, std::memory_order_relaxed); }
    const unsigned wpg = f->waves_per_group, threads = wpg * 64u;
    const uint64_t waves = nblocks * f->segments;
    const unsigned grid = (unsigned)((waves + wpg - 1) / wpg);
    // start/stop are attached to this dispatch itself (its begin/end timestamps)
    return hipExtModuleLaunchKernel(f->fn, grid * threads, 1, 1, threads, 1, 1, 0, st, args, nullptr, start, stop, 0);
}

uint64_t jit_small_launches() { return g_small_launches.load(std::memory_order_relaxed); }

void jit_destroy(JitFilter *f) {
    if (!f) return;
    if (f->small_thread.joinable()) f->small_thread.join();
    if (f->small) jit_destroy(f->small);
    if (f->module) (void)hipModuleUnload(f->module);
    delete f;
}

} // namespace ipcr
