//go:build hip

// Package hipengine implements pipeline.StreamingCompiledSimulator (internal/pipeline/sim.go:11-39) on the MI355X
// library libipcr_hip.so (C ABI: include/ipcr_hip.h).  Drop this directory into ipcr's internal/, point the cgo
// flags below at the library, apply appcore_core.patch and build with `CGO_ENABLED=1 go build -tags hip ./cmd/...`.
//
// This file is shipped as source: the image it was written in has no Go toolchain.  tests/test_go_shim_lint.py
// checks every C.ipcr_* / C.IPCR_* identifier and every struct field used here against include/ipcr_hip.h.
package hipengine

/*
#cgo CFLAGS: -I${SRCDIR}/../../third_party/ipcr_hip/include
#cgo LDFLAGS: -L${SRCDIR}/../../third_party/ipcr_hip -lipcr_hip -Wl,-rpath,${SRCDIR}/../../third_party/ipcr_hip
#include <stdlib.h>
#include "ipcr_hip.h"
*/
import "C"

import (
	"fmt"
	"runtime"
	"sync"
	"sync/atomic"
	"unsafe"

	"ipcr-core/engine"
	"ipcr-core/primer"
)

// Engine satisfies pipeline.Simulator, CompiledSimulator, ScratchCompiledSimulator and StreamingCompiledSimulator.
type Engine struct {
	cfg     engine.Config
	devices []int        // GPUs the workers are spread over (worker i -> devices[i % len(devices)])
	next    atomic.Int64 // scratches handed out so far

	mu     sync.Mutex
	panels map[*engine.CompiledPanel]*C.ipcr_panel // device handle per compiled panel
	scr    map[*engine.SimulationScratch]*C.ipcr_scratch

	ref       *engine.Engine                 // the reference engine, for panels outside the device path's limits
	refPanels map[*engine.CompiledPanel]bool // panels compiled by it
}

// New: devices = the GPUs to use (none given: all that ipcr_device_count reports).  One process, every GPU of the node, no
// collective: the chunks the pipeline hands to its workers are independent (internal/pipeline/pipeline.go:60-125), so
// worker i simply owns a scratch on device i mod N.
func New(c engine.Config, devices ...int) *Engine {
	if len(devices) == 0 {
		for d := 0; d < int(C.ipcr_device_count()); d++ {
			devices = append(devices, d)
		}
	}
	return &Engine{cfg: c, devices: devices, panels: map[*engine.CompiledPanel]*C.ipcr_panel{},
		scr: map[*engine.SimulationScratch]*C.ipcr_scratch{}, refPanels: map[*engine.CompiledPanel]bool{}}
}

func tooLong(pairs []primer.Pair, max int) bool {
	for _, p := range pairs {
		if len(p.Forward) > max || len(p.Reverse) > max {
			return true
		}
	}
	return false
}

func lastErr() error { return fmt.Errorf("ipcr_hip: %s", C.GoString(C.ipcr_last_error())) }

// CompilePanel: the exported fields (Pairs, Cfg) are what the pipeline reads (core/engine/compiled.go:77-80); the
// device handle is kept beside it.
func (e *Engine) CompilePanel(pairs []primer.Pair) *engine.CompiledPanel {
	// What the device path does not take goes to the reference engine unchanged (same interfaces, so the pipeline does
	// not notice): primers longer than IPCR_MAX_PRIMER_LEN (the reference allows 65 535, compiled.go:44-49), more than
	// IPCR_MAX_MM mismatches, or a negative --mismatches, which the library refuses (IPCR_ERR_INVALID) because the
	// reference's two matchers disagree about it (ac.go:207 vs match.go:79).
	if e.cfg.MaxMM < 0 || e.cfg.MaxMM > C.IPCR_MAX_MM || tooLong(pairs, C.IPCR_MAX_PRIMER_LEN) {
		if e.ref == nil {
			e.ref = engine.New(e.cfg)
		}
		cp := e.ref.CompilePanel(pairs)
		e.mu.Lock()
		e.refPanels[cp] = true
		e.mu.Unlock()
		return cp
	}
	cp := &engine.CompiledPanel{Pairs: append([]primer.Pair(nil), pairs...), Cfg: e.cfg}
	cfg := C.ipcr_config{max_mm: C.int32_t(e.cfg.MaxMM), terminal_window: C.int32_t(e.cfg.TerminalWindow),
		min_len: C.int32_t(e.cfg.MinLen), max_len: C.int32_t(e.cfg.MaxLen), hit_cap: C.int32_t(e.cfg.HitCap),
		seed_len: C.int32_t(e.cfg.SeedLen)}
	if e.cfg.Circular {
		cfg.circular = 1
	}
	if e.cfg.NeedSites {
		cfg.need_sites = 1 // (presentation only: the sites are sliced here, fillSites)
	}
	cps := make([]C.ipcr_pair, len(pairs))
	var frees []unsafe.Pointer
	for i, p := range pairs {
		id, f, r := C.CString(p.ID), C.CString(p.Forward), C.CString(p.Reverse)
		frees = append(frees, unsafe.Pointer(id), unsafe.Pointer(f), unsafe.Pointer(r))
		cps[i] = C.ipcr_pair{id: id, forward: f, reverse: r, min_product: C.int32_t(p.MinProduct), max_product: C.int32_t(p.MaxProduct)}
	}
	defer func() {
		for _, p := range frees {
			C.free(p)
		}
	}()
	var h *C.ipcr_panel
	var first *C.ipcr_pair
	if len(cps) > 0 {
		first = &cps[0]
	}
	if C.ipcr_panel_create(&cfg, first, C.int32_t(len(cps)), &h) != C.IPCR_OK {
		panic(lastErr()) // the reference panics on invalid panels too (core/primer/rc.go:27-34)
	}
	e.mu.Lock()
	e.panels[cp] = h
	e.mu.Unlock()
	runtime.SetFinalizer(cp, func(cp *engine.CompiledPanel) {
		e.mu.Lock()
		C.ipcr_panel_destroy(e.panels[cp])
		delete(e.panels, cp)
		e.mu.Unlock()
	})
	return cp
}

func (e *Engine) isRef(cp *engine.CompiledPanel) bool {
	e.mu.Lock()
	defer e.mu.Unlock()
	return e.refPanels[cp]
}

// NewSimulationScratch: the pipeline creates one per worker goroutine (pipeline.go:66-69): worker i -> device i mod N.
// The goroutine never selects a device itself (the Go runtime moves it between threads anyway): every library call
// selects the device of the scratch it is given and puts the thread's previous device back.
func (e *Engine) NewSimulationScratch(cp *engine.CompiledPanel) *engine.SimulationScratch {
	if e.isRef(cp) {
		return e.ref.NewSimulationScratch(cp)
	}
	s := engine.NewSimulationScratch(cp) // opaque token for the pipeline; device state is ours
	if len(e.devices) == 0 {
		panic("ipcr_hip: no HIP device (the scan path has no CPU fallback)")
	}
	e.mu.Lock()
	p := e.panels[cp]
	e.mu.Unlock()
	dev := e.devices[int(e.next.Add(1)-1)%len(e.devices)]
	var h *C.ipcr_scratch
	if C.ipcr_scratch_create_on(p, C.int32_t(dev), &h) != C.IPCR_OK {
		panic(lastErr())
	}
	e.mu.Lock()
	e.scr[s] = h
	e.mu.Unlock()
	runtime.SetFinalizer(s, func(s *engine.SimulationScratch) {
		e.mu.Lock()
		C.ipcr_scratch_destroy(e.scr[s])
		delete(e.scr, s)
		e.mu.Unlock()
	})
	return s
}

// Handles returns the device handles behind a compiled panel and a worker's scratch (nil, nil for a panel that went to
// the reference engine).  For packages that add device work to a worker's scan: hipprobe.
func (e *Engine) Handles(cp *engine.CompiledPanel, scratch *engine.SimulationScratch) (unsafe.Pointer, unsafe.Pointer) {
	e.mu.Lock()
	defer e.mu.Unlock()
	if e.refPanels[cp] {
		return nil, nil
	}
	return unsafe.Pointer(e.panels[cp]), unsafe.Pointer(e.scr[scratch])
}

// Reference returns the reference engine when `cp` was compiled by it (a panel outside the device path's limits).
func (e *Engine) Reference(cp *engine.CompiledPanel) *engine.Engine {
	if e.isRef(cp) {
		return e.ref
	}
	return nil
}

// ScanChunk runs ipcr_scan_chunk and converts the products; `after`, when not nil, is called with the scratch handle
// after the scan and before the first emit (the chunk's tiles are still in the scratch: hipprobe's batched rescan).
func (e *Engine) ScanChunk(seqID string, seq []byte, cp *engine.CompiledPanel, scratch *engine.SimulationScratch,
	after func(scratch unsafe.Pointer, n int) error, emit func(i int, p engine.Product) error) error {
	if scratch == nil {
		scratch = e.NewSimulationScratch(cp)
	}
	e.mu.Lock()
	p, s := e.panels[cp], e.scr[scratch]
	e.mu.Unlock()
	var ptr *C.uint8_t
	if len(seq) > 0 {
		ptr = (*C.uint8_t)(unsafe.Pointer(&seq[0])) // read-only, not retained after the call (cgo rule)
	}
	if C.ipcr_scan_chunk(p, s, ptr, C.uint64_t(len(seq)), nil, nil) != C.IPCR_OK {
		panic(lastErr())
	}
	var prods *C.ipcr_product
	var n C.int64_t
	if C.ipcr_scratch_products(s, &prods, &n) != C.IPCR_OK {
		panic(lastErr())
	}
	if after != nil {
		if err := after(unsafe.Pointer(s), int(n)); err != nil {
			return err
		}
	}
	for i, cpr := range unsafe.Slice(prods, int(n)) {
		pair := cp.Pairs[int(cpr.pair)]
		pr := engine.Product{ExperimentID: pair.ID, SequenceID: seqID,
			Start: int(cpr.start), End: int(cpr.end), Length: int(cpr.length),
			Type: "forward", FwdMM: int(cpr.fwd_mm), RevMM: int(cpr.rev_mm),
			FwdPrimer: pair.Forward, RevPrimer: pair.Reverse}
		revLen := len(pair.Reverse)
		if cpr._type == 1 {
			pr.Type, pr.FwdPrimer, pr.RevPrimer = "revcomp", pair.Reverse, pair.Forward
			revLen = len(pair.Forward)
		} else if pr.Start > pr.End {
			pr.RevPrimer = "" // the reference leaves it out of a forward wrap-around product (core/engine/engine.go:246-260)
		}
		pr.FwdMismatchIdx = idx(cpr.fwd_idx[:], int(cpr.n_fwd_idx)) // fresh slices: they cross goroutines
		pr.RevMismatchIdx = idx(cpr.rev_idx[:], int(cpr.n_rev_idx))
		if e.cfg.NeedSites {
			fillSites(&pr, seq, revLen)
		}
		if err := emit(i, pr); err != nil {
			return err
		}
	}
	return nil
}

// ForEachCompiledProduct -- core/engine/compiled.go:162-267 behind the C ABI.
func (e *Engine) ForEachCompiledProduct(seqID string, seq []byte, cp *engine.CompiledPanel,
	scratch *engine.SimulationScratch, emit func(engine.Product) error) error {
	if cp == nil || len(cp.Pairs) == 0 || emit == nil {
		return nil
	}
	if e.isRef(cp) { // a panel the device path does not take: the reference engine, unchanged
		return e.ref.ForEachCompiledProduct(seqID, seq, cp, scratch, emit)
	}
	return e.ScanChunk(seqID, seq, cp, scratch, nil, func(_ int, p engine.Product) error { return emit(p) })
}

// fillSites: FwdSite / RevSite as the join slices them in pretty mode (core/engine/engine.go:175-183, :242-249,
// :307-314, :372-379): the target under the left primer, and the reverse complement of the target under the right one.
func fillSites(pr *engine.Product, seq []byte, revLen int) {
	if flen := len(pr.FwdPrimer); pr.Start+flen <= len(seq) {
		pr.FwdSite = string(seq[pr.Start : pr.Start+flen])
	}
	if b := pr.End - revLen; b >= 0 && pr.End <= len(seq) {
		pr.RevSite = string(primer.RevComp(seq[b:pr.End])) // panics on a non-IUPAC byte, as the reference does (rc.go:27-34)
	}
}

func idx(a []C.uint8_t, n int) []int {
	if n == 0 {
		return nil
	}
	out := make([]int, n)
	for i := range out {
		out[i] = int(a[i])
	}
	return out
}

// The smaller interfaces delegate exactly as the reference's do (compiled.go:141-156, engine.go:33-51).
func (e *Engine) SimulateCompiledWithScratch(id string, seq []byte, cp *engine.CompiledPanel, s *engine.SimulationScratch) []engine.Product {
	var out []engine.Product
	_ = e.ForEachCompiledProduct(id, seq, cp, s, func(p engine.Product) error { out = append(out, p); return nil })
	return out
}

func (e *Engine) SimulateCompiled(id string, seq []byte, cp *engine.CompiledPanel) []engine.Product {
	return e.SimulateCompiledWithScratch(id, seq, cp, nil)
}

func (e *Engine) SimulateBatch(id string, seq []byte, pairs []primer.Pair) []engine.Product {
	return e.SimulateCompiled(id, seq, e.CompilePanel(pairs))
}
