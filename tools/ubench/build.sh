#!/bin/bash
# builds the development micro-benchmarks of this directory (run from anywhere; libipcr_hip.so must have been built)
set -e
cd "$(dirname "$0")"
hipcc -O2 -std=c++17 -D__HIP_PLATFORM_AMD__ pread_rate.cpp -o pread_rate -lpthread
hipcc -O2 -std=c++17 --offload-arch=gfx950 slab_pipeline.hip -o slab_pipeline -lpthread
hipcc -O2 -std=c++17 -mavx2 --offload-arch=gfx950 numa_probe.hip -o numa_probe -lpthread
hipcc -O2 -std=c++17 -I../../include -D__HIP_PLATFORM_AMD__ fasta_load.cpp -o fasta_load -L../../ipcr_amd -lipcr_hip -Wl,-rpath,'$ORIGIN/../../ipcr_amd' -Wl,-rpath,/opt/rocm/lib
[ -f valu_rates.hip ] && hipcc -O2 --offload-arch=gfx950 valu_rates.hip -o valu_rates || true
