//go:build hip

// Package hipprobe is ipcr-probe on the MI355X library: the probe rescan of every product
// (internal/visitors/probe.go:18-33 -> core/probe/annotate.go:14-17 -> core/oligo/oligo.go:19-77) moves from the single
// collector goroutine onto the WORKER that scanned the chunk, as one batched device call per chunk
// (ipcr_probe_scratch_products: the chunk's tiles are still in the worker's scratch), and the visitor only formats.
//
// The annotation rides through the unmodified pipeline inside the product: engine.Product.Thermo.Probe
// (core/engine/product.go: ProbeThermoDetails{Found, Strand, Pos, MM}) is nil in an ipcr-probe run, is not touched by
// the pipeline (internal/pipeline/pipeline.go:80-95 fills Seq and SourceFile only) and is taken out again by Visit
// before the product reaches a writer.  --chunk-size stays in force exactly as in the reference
// (internal/probeapp/app.go:108): chunk IDs, overlap, de-duplication and coordinate rebasing are the pipeline's.
//
// Source only (no Go toolchain in the build image); linted against include/ipcr_hip.h by tests/test_go_shim_lint.py.
package hipprobe

/*
#cgo CFLAGS: -I${SRCDIR}/../../third_party/ipcr_hip/include
#cgo LDFLAGS: -L${SRCDIR}/../../third_party/ipcr_hip -lipcr_hip -Wl,-rpath,${SRCDIR}/../../third_party/ipcr_hip
#include <stdlib.h>
#include "ipcr_hip.h"
*/
import "C"

import (
	"fmt"
	"strings"
	"unsafe"

	"ipcr-core/engine"
	"ipcr-core/primer"
	"ipcr-core/probe"
	"ipcr/internal/hipengine"
	"ipcr/internal/probeoutput"
)

const model = "hipprobe" // marks a Thermo block that only carries the worker's annotation

// Engine is hipengine.Engine plus the probe: every interface of internal/pipeline/sim.go:11-39, the streaming one
// annotating as it goes.  Construct it where internal/appcore/core.go:108-117 constructs the engine (appcore_core.patch
// passes the probe through appcore.Options).
type Engine struct {
	*hipengine.Engine
	Probe string // 5'->3', as given on the command line
	MaxMM int    // --probe-max-mm
}

func New(c engine.Config, probeSeq string, maxMM int, devices ...int) *Engine {
	return &Engine{Engine: hipengine.New(c, devices...), Probe: probeSeq, MaxMM: maxMM}
}

func (e *Engine) ForEachCompiledProduct(seqID string, seq []byte, cp *engine.CompiledPanel,
	scratch *engine.SimulationScratch, emit func(engine.Product) error) error {
	if cp == nil || len(cp.Pairs) == 0 || emit == nil {
		return nil
	}
	if ref := e.Reference(cp); ref != nil || strings.TrimSpace(e.Probe) == "" {
		return e.Engine.ForEachCompiledProduct(seqID, seq, cp, scratch, emit) // Visit annotates from p.Seq as the reference does
	}
	cprobe := C.CString(e.Probe)
	defer C.free(unsafe.Pointer(cprobe))
	var hits []C.ipcr_probe_hit
	after := func(s unsafe.Pointer, n int) error {
		if n == 0 {
			return nil
		}
		hits = make([]C.ipcr_probe_hit, n)
		if C.ipcr_probe_scratch_products((*C.ipcr_scratch)(s), cprobe, C.int32_t(e.MaxMM), &hits[0], C.int64_t(n)) != C.IPCR_OK {
			return fmt.Errorf("ipcr_hip: %s", C.GoString(C.ipcr_last_error()))
		}
		return nil
	}
	return e.ScanChunk(seqID, seq, cp, scratch, after, func(i int, p engine.Product) error {
		h := hits[i]
		ann := &engine.ProbeThermoDetails{Found: h.found != 0}
		if ann.Found {
			ann.Strand, ann.Pos, ann.MM = string(rune(h.strand)), int(h.pos), int(h.mm)
		}
		p.Thermo = &engine.ThermoDetails{Model: model, Probe: ann}
		return emit(p)
	})
}

// The slice-returning forms must annotate too (the pipeline uses them when a wrapper hides the streaming interface).
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

// Visitor replaces visitors.Probe (internal/visitors/probe.go:11-33): same fields, same output; it takes the worker's
// annotation when the product carries one and computes it from p.Seq as the reference does when it does not (a panel
// that went to the reference engine; a product that did not come through hipprobe.Engine).
type Visitor struct {
	Name    string
	Seq     string // 5'->3'
	MaxMM   int
	Require bool
}

func (v Visitor) Visit(p engine.Product) (bool, probeoutput.AnnotatedProduct, error) {
	var ann probe.Annotation
	if p.Thermo != nil && p.Thermo.Model == model && p.Thermo.Probe != nil {
		t := p.Thermo.Probe
		p.Thermo = nil // the writers must see the product as the reference would hand it to them
		ann = probe.Annotation{Found: t.Found, Strand: t.Strand, Pos: t.Pos, MM: t.MM}
		if ann.Found { // oligo.BestHit's Site: the amplicon under the hit, upper-cased (core/oligo/oligo.go:20,36,49-52)
			if end := ann.Pos + sitelen(v.Seq); end <= len(p.Seq) {
				ann.Site = strings.ToUpper(p.Seq[ann.Pos:end])
			}
		}
	} else {
		ann = probe.AnnotateAmplicon(p.Seq, v.Seq, v.MaxMM)
	}
	if v.Require && !ann.Found {
		return false, probeoutput.AnnotatedProduct{}, nil
	}
	return true, probeoutput.AnnotatedProduct{
		Product:     p,
		ProbeName:   v.Name,
		ProbeSeq:    strings.ToUpper(v.Seq),
		ProbeFound:  ann.Found,
		ProbeStrand: ann.Strand,
		ProbePos:    ann.Pos,
		ProbeMM:     ann.MM,
		ProbeSite:   ann.Site,
	}, nil
}

// length of the probe after oligo.Validate's normalisation (white space and quotes dropped, core/oligo/validate.go)
func sitelen(raw string) int {
	n := 0
	for _, ch := range raw {
		switch ch {
		case ' ', '\t', '\n', '\r', '\v', '\f', '\'', '"':
		default:
			n++
		}
	}
	return n
}
