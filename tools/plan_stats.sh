#!/bin/bash
# stored wires per unit kind of the production instantiation (host only, no GPU): tools/plan_stats.cpp against the HIP-on-fibers shim
cd "$(dirname "$0")/.." && /opt/rocm/lib/llvm/bin/clang++ -x c++ -std=c++17 -O1 -DPOB_HOSTSIM -D__HIPCC__ -I tests/hostsim -I proof_of_burn_amd/csrc -Wno-unused-value -Wno-unknown-attributes \
  -Wno-ignored-attributes -Wno-undefined-inline -fbracket-depth=1024 tools/plan_stats.cpp tests/hostsim/hostsim_rt.cpp -o /tmp/plan_stats && /tmp/plan_stats
